// Masked multi-head self-attention, forward and backward (unirec/model/modules.py:284-311 with the
// additive mask of unirec/model/sequential/sasrec.py:40-57).
//
// SASRec heads are tiny (n_heads=16 => head dim 4..8), so QK^T and P.V run on the VALU and MFMA is
// reserved for the dense projections.  Work decomposition, CDNA-style:
//   * one WAVE owns one (sequence, head, 64-row chunk); one LANE owns one query row (row passes) or one
//     key row (column pass), so every softmax reduction is lane-local -- no cross-lane traffic at all;
//   * the operand that all 64 lanes share in an iteration (K_j / V_j in the row passes, Q_i / dO_i in the
//     column pass) is WAVE-UNIFORM: it is fetched with scalar loads (s_load_dwordx8 through the scalar
//     cache) straight into SGPRs and fed to v_fma as a scalar operand.  No LDS, no barriers, and the
//     vector memory pipe only carries each lane's own row once.
// Nothing of size [B,h,L,L] is ever written: the backward recomputes P from Q, K and the saved
// log-sum-exp.
//
// Mask semantics are the reference's: allowed(i,j) = item_seq[j] > 0 and (j <= i if causal); masked keys
// get -10000 added, i.e. softmax weight exp(-10000 - max) == 0 exactly in fp32, so they are skipped.
//   * A row with NO allowed key while the sequence has valid keys (the left-padding prefix in causal mode)
//     is a row whose output can never reach the loss: it is only ever read as a key/value at a padded
//     position (masked) or as the query of another such row (verified on the reference:
//     max |delta| == 0).  Those rows are written as zeros and contribute nothing to the backward.
//   * A sequence with no valid key at all (empty history) takes the literal path: every key gets
//     s/sqrt(hd) + (-10000.0f) and the softmax runs over all L keys, exactly as torch evaluates it.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace ur {

struct AttnDims {
  int B, L, d, H, hd, causal, nchunk;
  float scale;    // 1/sqrt(hd)
  float sqrt_hd;  // sqrt(hd) for the literal (division) path
  const int* seq_base;   // compacted rows (nullable): position l of sequence b is row seq_base[b] + l, for l >= seq_pad[b]
  const int* seq_pad;
  unsigned dkey, dthresh;   // dropout on the probabilities (dthresh == 0: off): row id (b*H + h)*L + i, column j
  float dscale;
};
__device__ __forceinline__ unsigned attn_rowkey(const AttnDims& p, int b, int h, int i) {
  return mix32((unsigned)((b * p.H + h) * p.L + i) ^ p.dkey);
}
// multiplier of P[i,j]; DROP is a compile-time switch: the kernels without dropout carry none of this
template <bool DROP>
__device__ __forceinline__ float attn_keep(const AttnDims& p, unsigned rowkey, int j) {
  if constexpr (DROP) return drop_mul(rowkey, (unsigned)j, p.dthresh, p.dscale);
  else return 1.0f;
}
// first row of sequence b's (virtual) position 0 and the number of leading positions that have no row
__device__ __forceinline__ void seq_rows(const AttnDims& p, int b, long long& row0, int& pad) {
  row0 = p.seq_base ? (long long)p.seq_base[b] : (long long)b * p.L;
  pad = p.seq_base ? p.seq_pad[b] : 0;
}

// dot of a per-lane register row with a wave-uniform memory row (HD exact: no guards, so the compiler can
// fetch the row with wide scalar loads)
template <int HD>
__device__ __forceinline__ float dot_u(const float (&q)[HD], const float* __restrict__ k) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HD; ++c) s = fmaf(q[c], k[c], s);
  return s;
}

#define UR_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

// One (sequence, head) per workgroup, 1-D grid.  A head's slice of a qkv / ctx row is HD * 4 bytes, so 128 / (HD * 4) heads
// share every 128-byte line; workgroup n runs on XCD n % 8 (each XCD has its own L2), so those heads are given CONSECUTIVE
// slots of ONE XCD: the line is fetched into one L2 once instead of once per head, and the partial-line stores merge there.
__host__ __device__ inline int attn_heads_per_line(int hd) { return hd >= 32 ? 1 : 32 / hd; }
__host__ inline unsigned attn_bh_grid(int B, int H, int hd) {
  const int hpl = attn_heads_per_line(hd), G = (H + hpl - 1) / hpl;
  return 8u * hpl * (unsigned)(((long long)B * G + 7) / 8);
}
__device__ __forceinline__ bool attn_bh_of_block(const AttnDims& p, int hd, int& b, int& h) {
  const int hpl = attn_heads_per_line(hd), G = (p.H + hpl - 1) / hpl;
  const int n = blockIdx.x, t = n >> 3;
  const int q = 8 * (t / hpl) + (n & 7);
  b = q / G;
  h = (q % G) * hpl + t % hpl;
  return b < p.B && h < p.H;
}

// index of the first key with item_seq > 0, or L when the sequence is all padding (wave-uniform result)
__device__ __forceinline__ int first_valid_key(const int* __restrict__ sq, int L, int lane) {
  for (int j0 = 0; j0 < L; j0 += 64) {
    const unsigned long long m = __ballot(j0 + lane < L && sq[j0 + lane] > 0);
    if (m) return j0 + (int)__builtin_ctzll(m);
  }
  return L;
}

template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ seq, AttnDims p,
                                                       float* __restrict__ ctx, float* __restrict__ lse) {
  const int lane = threadIdx.x & 63;
  const int item = UR_UNIFORM((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
  const int h = item / p.nchunk, ck = item % p.nchunk;
  if (h >= p.H) return;
  const int b = blockIdx.x, L = p.L, ld = 3 * p.d;
  const int i = ck * 64 + lane;
  const bool active = i < L;
  const int ii = active ? i : L - 1;
  const float* __restrict__ base = qkv + (long long)b * L * ld;
  const int* __restrict__ sq = seq + (long long)b * L;
  float q[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) q[c] = base[(long long)ii * ld + h * HD + c];
  const float* __restrict__ Kb = base + p.d + h * HD;
  const float* __restrict__ Vb = base + 2 * p.d + h * HD;
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const unsigned rk = attn_rowkey(p, b, h, ii);
  float m = -INFINITY, l = 0.f, o[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) o[c] = 0.f;
  if (fv < L) {
    const int jend = p.causal ? min(L, ck * 64 + 64) : L;
    // interior zeros ('unorder' masking) are predicated, not branched, so the loops unroll and the scalar
    // loads of several keys are in flight together
#pragma unroll 4
    for (int j = fv; j < jend; ++j) {
      const float s = dot_u<HD>(q, Kb + (long long)j * ld) * p.scale;
      if (sq[j] > 0 && (!p.causal || j <= i)) m = fmaxf(m, s);
    }
#pragma unroll 4
    for (int j = fv; j < jend; ++j) {
      const float s = dot_u<HD>(q, Kb + (long long)j * ld) * p.scale;
      const float pj = (sq[j] > 0 && (!p.causal || j <= i)) ? __expf(s - m) : 0.f;
      l += pj;
      const float pd = pj * attn_keep<DROP>(p, rk, j);
      const float* __restrict__ vr = Vb + (long long)j * ld;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = fmaf(pd, vr[c], o[c]);
    }
  } else {  // empty history: literal path over all L keys
    for (int j = 0; j < L; ++j) m = fmaxf(m, dot_u<HD>(q, Kb + (long long)j * ld) / p.sqrt_hd + -10000.0f);
    for (int j = 0; j < L; ++j) {
      const float pj = __expf((dot_u<HD>(q, Kb + (long long)j * ld) / p.sqrt_hd + -10000.0f) - m);
      l += pj;
      const float pd = pj * attn_keep<DROP>(p, rk, j);
      const float* __restrict__ vr = Vb + (long long)j * ld;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = fmaf(pd, vr[c], o[c]);
    }
  }
  if (!active) return;
  const bool dead = l == 0.f;   // padded-prefix row of a non-empty sequence: unreachable from the loss
  const float inv_l = dead ? 0.f : 1.0f / l;
  float* out = ctx + ((long long)b * L + i) * p.d + h * HD;
#pragma unroll
  for (int c = 0; c < HD; ++c) out[c] = o[c] * inv_l;
  lse[((long long)b * p.H + h) * L + i] = dead ? 0.f : m + __logf(l);
}

// Backward prep: Dd[b,h,i] = sum_c dO[b,i,h,c] * O[b,i,h,c]
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                            AttnDims p, float* __restrict__ Dd) {
  const int b = blockIdx.x, L = p.L;
  for (int idx = threadIdx.x; idx < p.H * L; idx += blockDim.x) {
    const int i = idx / p.H, h = idx % p.H;   // consecutive threads -> consecutive heads of one row: coalesced
    const float* o = ctx + ((long long)b * L + i) * p.d + h * p.hd;
    const float* g = dctx + ((long long)b * L + i) * p.d + h * p.hd;
    float s = 0.f;
    for (int c = 0; c < p.hd; ++c) s = fmaf(o[c], g[c], s);
    Dd[((long long)b * p.H + h) * L + i] = s;
  }
}

// Backward. dqkv[:, 0:d] = dQ, [d:2d] = dK, [2d:3d] = dV.  Row pass then column pass, same wave, no LDS.
template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const int* __restrict__ seq,
                                                       const float* __restrict__ dctx, const float* __restrict__ lse,
                                                       const float* __restrict__ Dd, AttnDims p, float* __restrict__ dqkv) {
  const int lane = threadIdx.x & 63;
  const int item = UR_UNIFORM((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
  const int h = item / p.nchunk, ck = item % p.nchunk;
  if (h >= p.H) return;
  const int b = blockIdx.x, L = p.L, ld = 3 * p.d;
  const int r = ck * 64 + lane;         // this lane's row: query row in the row pass, key row in the column pass
  const bool active = r < L;
  const int rr = active ? r : L - 1;
  const float* __restrict__ base = qkv + (long long)b * L * ld;
  const float* __restrict__ gbase = dctx + (long long)b * L * p.d + h * HD;
  const int* __restrict__ sq = seq + (long long)b * L;
  const float* __restrict__ lse_h = lse + ((long long)b * p.H + h) * L;
  const float* __restrict__ D_h = Dd + ((long long)b * p.H + h) * L;
  const float* __restrict__ Qb = base + h * HD;
  const float* __restrict__ Kb = base + p.d + h * HD;
  const float* __restrict__ Vb = base + 2 * p.d + h * HD;
  float* orow = dqkv + ((long long)b * L + rr) * ld + h * HD;
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const bool literal = fv >= L;                    // empty history (wave-uniform)
  const int dead_below = p.causal ? fv : 0;        // rows i < dead_below have no allowed key and are unreachable

  {  // ---- row pass: dQ_i = scale * sum_j dS_ij K_j,  dS_ij = P_ij (dO_i . V_j - D_i)
    float q[HD], g[HD], dq[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      q[c] = Qb[(long long)rr * ld + c];
      g[c] = gbase[(long long)rr * p.d + c];
      dq[c] = 0.f;
    }
    const float li = lse_h[rr], Di = D_h[rr];
    const unsigned rk = attn_rowkey(p, b, h, rr);
    if (!literal) {
      const bool live = r >= dead_below;
      const int jend = p.causal ? min(L, ck * 64 + 64) : L;
#pragma unroll 4
      for (int j = fv; j < jend; ++j) {
        const float* __restrict__ kr = Kb + (long long)j * ld;
        const float s = dot_u<HD>(q, kr) * p.scale;
        const float pj = (live && sq[j] > 0 && (!p.causal || j <= r)) ? __expf(s - li) : 0.f;
        const float ds = pj * (attn_keep<DROP>(p, rk, j) * dot_u<HD>(g, Vb + (long long)j * ld) - Di);
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, kr[c], dq[c]);
      }
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] *= p.scale;
    } else {
      for (int j = 0; j < L; ++j) {
        const float* __restrict__ kr = Kb + (long long)j * ld;
        const float pj = __expf((dot_u<HD>(q, kr) / p.sqrt_hd + -10000.0f) - li);
        const float ds = pj * (attn_keep<DROP>(p, rk, j) * dot_u<HD>(g, Vb + (long long)j * ld) - Di);
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, kr[c], dq[c]);
      }
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] /= p.sqrt_hd;
    }
    if (active) {
#pragma unroll
      for (int c = 0; c < HD; ++c) orow[c] = dq[c];
    }
  }
  {  // ---- column pass: dK_j = sum_i dS_ij Q_i * scale,  dV_j = sum_i P_ij dO_i
    float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      k[c] = Kb[(long long)rr * ld + c];
      v[c] = Vb[(long long)rr * ld + c];
      dk[c] = 0.f;
      dv[c] = 0.f;
    }
    if (!literal) {
      const bool vj = sq[rr] > 0;
      const int ibeg = max(dead_below, p.causal ? ck * 64 : 0);  // earlier rows see none of this chunk's keys
#pragma unroll 4
      for (int i = ibeg; i < L; ++i) {
        const float* __restrict__ qr = Qb + (long long)i * ld;
        const float* __restrict__ gr = gbase + (long long)i * p.d;
        const float s = dot_u<HD>(k, qr) * p.scale;
        const float pj = (vj && (!p.causal || r <= i)) ? __expf(s - lse_h[i]) : 0.f;
        const float mk = attn_keep<DROP>(p, attn_rowkey(p, b, h, i), rr);
        const float ds = pj * (mk * dot_u<HD>(v, gr) - D_h[i]) * p.scale;
        const float pd = pj * mk;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          dk[c] = fmaf(ds, qr[c], dk[c]);
          dv[c] = fmaf(pd, gr[c], dv[c]);
        }
      }
    } else {
      for (int i = 0; i < L; ++i) {
        const float* __restrict__ qr = Qb + (long long)i * ld;
        const float* __restrict__ gr = gbase + (long long)i * p.d;
        const float pj = __expf((dot_u<HD>(k, qr) / p.sqrt_hd + -10000.0f) - lse_h[i]);
        const float mk = attn_keep<DROP>(p, attn_rowkey(p, b, h, i), rr);
        const float ds = pj * (mk * dot_u<HD>(v, gr) - D_h[i]) / p.sqrt_hd;
        const float pd = pj * mk;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          dk[c] = fmaf(ds, qr[c], dk[c]);
          dv[c] = fmaf(pd, gr[c], dv[c]);
        }
      }
    }
    if (active) {
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        orow[p.d + c] = dk[c];
        orow[2 * p.d + c] = dv[c];
      }
    }
  }
}

// HD contiguous floats <-> registers with 16-byte accesses when the head dim allows (rows are 16-B aligned: d % 4 == 0)
template <int HD>
__device__ __forceinline__ void load_vec(const float* __restrict__ src, float (&dst)[HD]) {
  if constexpr (HD % 4 == 0) {
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 v = *(const float4*)(src + c);
      dst[c] = v.x; dst[c + 1] = v.y; dst[c + 2] = v.z; dst[c + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int c = 0; c < HD; ++c) dst[c] = src[c];
  }
}
template <int HD>
__device__ __forceinline__ void store_vec(float* __restrict__ dst, const float (&src)[HD]) {
  if constexpr (HD % 4 == 0) {
#pragma unroll
    for (int c = 0; c < HD; c += 4) *(float4*)(dst + c) = make_float4(src[c], src[c + 1], src[c + 2], src[c + 3]);
  } else {
#pragma unroll
    for (int c = 0; c < HD; ++c) dst[c] = src[c];
  }
}

// ------------------------------------------------------------------------------------------------------------
// Last-row specialisation: in the final layer only the query at position L-1 can reach the loss,
// so attention degenerates to ONE query per (sequence, head).  Here one wave = one (sequence, head), lanes = keys:
// the softmax max / sum and the P.V contraction are wavefront xor-shuffle reductions.
template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_last_fwd_kernel(const float* __restrict__ q_last, const float* __restrict__ qkv,
                                                            const int* __restrict__ seq, AttnDims p, float* __restrict__ ctx_last,
                                                            float* __restrict__ lse_last, AttnQProj qp) {
  const int lane = threadIdx.x & 63;
  const int h = UR_UNIFORM((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
  if (h >= p.H) return;
  const int b = blockIdx.x, L = p.L, ld = 3 * p.d;
  long long row0;
  int pad;
  seq_rows(p, b, row0, pad);
  const float* __restrict__ base = qkv + row0 * ld;
  const int* __restrict__ sq = seq + (long long)b * L;
  float qr[HD];
  if (qp.wq != nullptr) {   // project this head's query: lane = input feature(s), one wave reduction per output
    const long long xr = qp.xrow ? (long long)qp.xrow[b] : (long long)b * qp.xstride + qp.xoff;
    const float* __restrict__ xrow = qp.x + xr * p.d;
#pragma unroll
    for (int c = 0; c < HD; ++c) qr[c] = 0.f;
    for (int k = lane; k < p.d; k += 64) {
      const float xv = xrow[k];
#pragma unroll
      for (int c = 0; c < HD; ++c) qr[c] = fmaf(xv, qp.wq[(long long)(h * HD + c) * p.d + k], qr[c]);
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) qr[c] = wave_sum(qr[c]) + qp.bq[h * HD + c];
#pragma unroll
    for (int c = 0; c < HD; ++c)
      if (lane == c) qp.q_out[(long long)b * p.d + h * HD + c] = qr[c];
    if (qp.x_out != nullptr && lane < HD) qp.x_out[(long long)b * p.d + h * HD + lane] = xrow[h * HD + lane];
  } else {
    const float* __restrict__ qsrc = q_last + (long long)b * p.d + h * HD;   // wave-uniform row
#pragma unroll
    for (int c = 0; c < HD; ++c) qr[c] = qsrc[c];
  }
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const bool literal = fv >= L;
  const unsigned rk = attn_rowkey(p, b, h, L - 1);
  float m = -INFINITY, l = 0.f, o[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) o[c] = 0.f;
  for (int j0 = 0; j0 < L; j0 += 64) {
    const int j = j0 + lane;
    const bool in = j < L;
    const int jj = max(in ? j : L - 1, pad);
    float kr[HD], vr[HD];
    load_vec<HD>(base + (long long)jj * ld + p.d + h * HD, kr);
    load_vec<HD>(base + (long long)jj * ld + 2 * p.d + h * HD, vr);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) s = fmaf(qr[c], kr[c], s);
    const bool allowed = in && (literal || sq[min(j, L - 1)] > 0);   // the mask is indexed by position, not by (clamped) row
    const float sv = allowed ? (literal ? s / p.sqrt_hd + -10000.0f : s * p.scale) : -INFINITY;
    const float mn = fmaxf(m, group_max<64>(sv));
    if (mn == -INFINITY) continue;   // nothing allowed so far (uniform)
    const float corr = __expf(m - mn);
    const float pj = allowed ? __expf(sv - mn) : 0.f;
    l = l * corr + wave_sum(pj);
    const float pd = pj * attn_keep<DROP>(p, rk, j);
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = o[c] * corr + wave_sum(pd * vr[c]);
    m = mn;
  }
  const float inv_l = 1.0f / l;
#pragma unroll
  for (int c = 0; c < HD; ++c)
    if (lane == c) ctx_last[(long long)b * p.d + h * HD + c] = o[c] * inv_l;
  if (lane == 0) lse_last[(long long)b * p.H + h] = m + __logf(l);
}

// dq_last [B,d]; dK, dV written into dqkv[:, d:3d] for ALL rows (zeros where the key is masked); dqkv[:, 0:d] untouched.
template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_last_bwd_kernel(const float* __restrict__ q_last, const float* __restrict__ qkv,
                                                            const int* __restrict__ seq, const float* __restrict__ ctx_last,
                                                            const float* __restrict__ dctx_last, const float* __restrict__ lse_last,
                                                            AttnDims p, float* __restrict__ dq_last, float* __restrict__ dqkv) {
  const int lane = threadIdx.x & 63;
  const int h = UR_UNIFORM((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
  if (h >= p.H) return;
  const int b = blockIdx.x, L = p.L, ld = 3 * p.d;
  long long row0;
  int pad;
  seq_rows(p, b, row0, pad);
  const float* __restrict__ base = qkv + row0 * ld;
  const int* __restrict__ sq = seq + (long long)b * L;
  const float* __restrict__ qr = q_last + (long long)b * p.d + h * HD;
  const float* __restrict__ gr = dctx_last + (long long)b * p.d + h * HD;
  const float* __restrict__ orow = ctx_last + (long long)b * p.d + h * HD;
  const float ls = lse_last[(long long)b * p.H + h];
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const bool literal = fv >= L;
  const float f = literal ? 1.0f / p.sqrt_hd : p.scale;
  float D = 0.f;
#pragma unroll
  for (int c = 0; c < HD; ++c) D = fmaf(gr[c], orow[c], D);
  const unsigned rk = attn_rowkey(p, b, h, L - 1);
  float dq[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) dq[c] = 0.f;
  for (int j0 = 0; j0 < L; j0 += 64) {
    const int j = j0 + lane;
    const bool in = j < L;
    const int jj = max(in ? j : L - 1, pad);
    float kr[HD], vr[HD];
    load_vec<HD>(base + (long long)jj * ld + p.d + h * HD, kr);
    load_vec<HD>(base + (long long)jj * ld + 2 * p.d + h * HD, vr);
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      s = fmaf(qr[c], kr[c], s);
      dp = fmaf(gr[c], vr[c], dp);
    }
    const bool allowed = in && (literal || sq[min(j, L - 1)] > 0);   // the mask is indexed by position, not by (clamped) row
    const float sv = literal ? s / p.sqrt_hd + -10000.0f : s * p.scale;
    const float pj = allowed ? __expf(sv - ls) : 0.f;
    const float mk = attn_keep<DROP>(p, rk, j);
    const float ds = pj * (mk * dp - D) * f;
    if (in && j >= pad) {
      float* out = dqkv + (row0 + j) * ld + h * HD;
      float dkr[HD], dvr[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        dkr[c] = ds * qr[c];
        dvr[c] = pj * mk * gr[c];
      }
      store_vec<HD>(out + p.d, dkr);
      store_vec<HD>(out + 2 * p.d, dvr);
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) dq[c] += wave_sum(ds * kr[c]);
  }
#pragma unroll
  for (int c = 0; c < HD; ++c)
    if (lane == c) dq_last[(long long)b * p.d + h * HD + c] = dq[c];
}

// ------------------------------------------------------------------------------------------------------------
// Register-broadcast variants for small heads (head dim <= 16, SASRec's default 16 heads): the shared operand row is
// not re-fetched from memory at all.  Each lane keeps ITS OWN key row (K_j, V_j) -- or query row in the column
// pass -- in VGPRs, and iteration j broadcasts lane j's registers to the whole wave with v_readlane_b32 (the value
// lands in an SGPR and feeds v_fma as a scalar operand).  Per wave the only memory traffic is one row per lane per
// 64-key chunk; the scalar-cache refill traffic that bounded the s_load variants above is gone.
__device__ __forceinline__ float bcast(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_rl_kernel(const float* __restrict__ qkv, const int* __restrict__ seq, AttnDims p,
                                                          float* __restrict__ ctx, float* __restrict__ lse) {
  const int lane = threadIdx.x & 63;
  const int item = UR_UNIFORM((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
  const int h = item / p.nchunk, ck = item % p.nchunk;
  if (h >= p.H) return;
  const int b = blockIdx.x, L = p.L, ld = 3 * p.d;
  const int i = ck * 64 + lane;
  const bool active = i < L;
  const int ii = active ? i : L - 1;
  const float* __restrict__ base = qkv + (long long)b * L * ld;
  const int* __restrict__ sq = seq + (long long)b * L;
  float q[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) q[c] = base[(long long)ii * ld + h * HD + c];
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const unsigned rk = attn_rowkey(p, b, h, ii);
  float m = -INFINITY, l = 0.f, o[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) o[c] = 0.f;
  if (fv < L) {
    // scores are pre-scaled: q <- q / sqrt(hd), so s = q . k needs no multiply in the loop
#pragma unroll
    for (int c = 0; c < HD; ++c) q[c] *= p.scale;
    const int nkc = p.causal ? ck + 1 : p.nchunk;   // key chunks this query chunk can see
    const int c0 = fv >> 6;
    for (int pass = 0; pass < 2; ++pass) {
      for (int kc = c0; kc < nkc; ++kc) {
        const int jl = kc * 64 + lane, jc = min(jl, L - 1);
        float kreg[HD], vreg[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          kreg[c] = base[(long long)jc * ld + p.d + h * HD + c];
          vreg[c] = pass ? base[(long long)jc * ld + 2 * p.d + h * HD + c] : 0.f;
        }
        const unsigned long long vmask = __ballot(jl < L && sq[jc] > 0);
        const int jn = min(64, L - kc * 64);
        const int j0 = (kc == c0) ? (fv & 63) : 0;
        const int ilim = p.causal ? i - kc * 64 : 64;   // key jj is visible iff jj <= ilim
        // two keys per iteration: the two broadcast->fma chains interleave, hiding the VALU->SGPR->VALU wait states
        int jj = j0;
        for (; jj + 1 < jn; jj += 2) {
          float ka[HD], kb[HD];
#pragma unroll
          for (int c = 0; c < HD; ++c) { ka[c] = bcast(kreg[c], jj); kb[c] = bcast(kreg[c], jj + 1); }
          float sa = 0.f, sb = 0.f;
#pragma unroll
          for (int c = 0; c < HD; ++c) { sa = fmaf(q[c], ka[c], sa); sb = fmaf(q[c], kb[c], sb); }
          const bool oka = ((vmask >> jj) & 1ull) && jj <= ilim, okb = ((vmask >> (jj + 1)) & 1ull) && jj + 1 <= ilim;
          if (pass == 0) {
            m = fmaxf(m, fmaxf(oka ? sa : -INFINITY, okb ? sb : -INFINITY));
          } else {
            float pa = oka ? __expf(sa - m) : 0.f, pb = okb ? __expf(sb - m) : 0.f;
            l += pa + pb;
            if constexpr (DROP) { pa *= attn_keep<DROP>(p, rk, kc * 64 + jj); pb *= attn_keep<DROP>(p, rk, kc * 64 + jj + 1); }
            float va[HD], vb[HD];
#pragma unroll
            for (int c = 0; c < HD; ++c) { va[c] = bcast(vreg[c], jj); vb[c] = bcast(vreg[c], jj + 1); }
#pragma unroll
            for (int c = 0; c < HD; ++c) o[c] = fmaf(pb, vb[c], fmaf(pa, va[c], o[c]));
          }
        }
        if (jj < jn) {
          float sa = 0.f;
#pragma unroll
          for (int c = 0; c < HD; ++c) sa = fmaf(q[c], bcast(kreg[c], jj), sa);
          const bool oka = ((vmask >> jj) & 1ull) && jj <= ilim;
          if (pass == 0) {
            if (oka) m = fmaxf(m, sa);
          } else {
            float pa = oka ? __expf(sa - m) : 0.f;
            l += pa;
            pa *= attn_keep<DROP>(p, rk, kc * 64 + jj);
#pragma unroll
            for (int c = 0; c < HD; ++c) o[c] = fmaf(pa, bcast(vreg[c], jj), o[c]);
          }
        }
      }
    }
  } else {  // empty history: literal path over all L keys (rare; plain loops)
    const float* __restrict__ Kb = base + p.d + h * HD;
    const float* __restrict__ Vb = base + 2 * p.d + h * HD;
    for (int j = 0; j < L; ++j) m = fmaxf(m, dot_u<HD>(q, Kb + (long long)j * ld) / p.sqrt_hd + -10000.0f);
    for (int j = 0; j < L; ++j) {
      const float pj = __expf((dot_u<HD>(q, Kb + (long long)j * ld) / p.sqrt_hd + -10000.0f) - m);
      l += pj;
      const float pd = pj * attn_keep<DROP>(p, rk, j);
      const float* __restrict__ vr = Vb + (long long)j * ld;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = fmaf(pd, vr[c], o[c]);
    }
  }
  if (!active) return;
  const bool dead = l == 0.f;   // padded-prefix row of a non-empty sequence: unreachable from the loss
  const float inv_l = dead ? 0.f : 1.0f / l;
  float* out = ctx + ((long long)b * L + i) * p.d + h * HD;
#pragma unroll
  for (int c = 0; c < HD; ++c) out[c] = o[c] * inv_l;
  lse[((long long)b * p.H + h) * L + i] = dead ? 0.f : m + __logf(l);
}

template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_rl_kernel(const float* __restrict__ qkv, const int* __restrict__ seq,
                                                          const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                          const float* __restrict__ lse, AttnDims p, float* __restrict__ dqkv) {
  const int lane = threadIdx.x & 63;
  const int item = UR_UNIFORM((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
  const int h = item / p.nchunk, ck = item % p.nchunk;
  if (h >= p.H) return;
  const int b = blockIdx.x, L = p.L, ld = 3 * p.d;
  const int r = ck * 64 + lane;          // this lane's row: query row in the row pass, key row in the column pass
  const bool active = r < L;
  const int rr = active ? r : L - 1;
  const float* __restrict__ base = qkv + (long long)b * L * ld;
  const float* __restrict__ gbase = dctx + (long long)b * L * p.d + h * HD;
  const float* __restrict__ obase = ctx + (long long)b * L * p.d + h * HD;
  const int* __restrict__ sq = seq + (long long)b * L;
  const float* __restrict__ lse_h = lse + ((long long)b * p.H + h) * L;
  float* orow = dqkv + ((long long)b * L + rr) * ld + h * HD;
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const bool literal = fv >= L;
  const int dead_below = (p.causal && !literal) ? fv : 0;   // query rows below this have no allowed key
  const float f = literal ? 1.0f / p.sqrt_hd : p.scale;
  // own row as a query: q, dO, lse, D = dO . O
  float q[HD], g[HD];
  float Dr = 0.f;
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    q[c] = base[(long long)rr * ld + h * HD + c];
    g[c] = gbase[(long long)rr * p.d + c];
    Dr = fmaf(g[c], obase[(long long)rr * p.d + c], Dr);
  }
  const float lr = lse_h[rr];
  const unsigned rk = attn_rowkey(p, b, h, rr);
  {  // ---- row pass: dQ_i = f * sum_j dS_ij K_j
    float dq[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dq[c] = 0.f;
    const bool live = literal || r >= dead_below;
    const int nkc = (p.causal && !literal) ? ck + 1 : p.nchunk;
    const int c0 = literal ? 0 : fv >> 6;
    for (int kc = c0; kc < nkc; ++kc) {
      const int jl = kc * 64 + lane, jc = min(jl, L - 1);
      float kreg[HD], vreg[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        kreg[c] = base[(long long)jc * ld + p.d + h * HD + c];
        vreg[c] = base[(long long)jc * ld + 2 * p.d + h * HD + c];
      }
      const unsigned long long vmask = literal ? ~0ull : __ballot(jl < L && sq[jc] > 0);
      const int jn = min(64, L - kc * 64);
      const int j0 = (!literal && kc == c0) ? (fv & 63) : 0;
#pragma unroll 4
      for (int jj = j0; jj < jn; ++jj) {
        float kj[HD];
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          kj[c] = bcast(kreg[c], jj);
          s = fmaf(q[c], kj[c], s);
          dp = fmaf(g[c], bcast(vreg[c], jj), dp);
        }
        const int j = kc * 64 + jj;
        const bool ok = live && ((vmask >> jj) & 1ull) && (literal || !p.causal || j <= r);
        const float sv = literal ? s / p.sqrt_hd + -10000.0f : s * p.scale;
        const float ds = ok ? __expf(sv - lr) * (attn_keep<DROP>(p, rk, j) * dp - Dr) : 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, kj[c], dq[c]);
      }
    }
    if (active) {
#pragma unroll
      for (int c = 0; c < HD; ++c) orow[c] = dq[c] * f;
    }
  }
  {  // ---- column pass: dK_j = f * sum_i dS_ij Q_i ,  dV_j = sum_i P_ij dO_i   (own row as a key)
    float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      k[c] = base[(long long)rr * ld + p.d + h * HD + c];
      v[c] = base[(long long)rr * ld + 2 * p.d + h * HD + c];
      dk[c] = 0.f;
      dv[c] = 0.f;
    }
    const bool vj = literal || sq[rr] > 0;
    const int qc0 = (p.causal && !literal) ? max(ck, dead_below >> 6) : 0;   // earlier query chunks see none of these keys
    for (int qc = qc0; qc < p.nchunk; ++qc) {
      // lane t of the wave holds query row qc*64+t for broadcasting (for qc == ck that is this lane's own row)
      const int il = qc * 64 + lane, ic = min(il, L - 1);
      float qreg[HD], greg[HD];
      float Dq = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        qreg[c] = base[(long long)ic * ld + h * HD + c];
        greg[c] = gbase[(long long)ic * p.d + c];
        Dq = fmaf(greg[c], obase[(long long)ic * p.d + c], Dq);
      }
      const float lq = lse_h[ic];
      const int in = min(64, L - qc * 64);
      const int i0 = (qc == (dead_below >> 6)) ? (dead_below & 63) : 0;
#pragma unroll 4
      for (int it = i0; it < in; ++it) {
        float qi[HD], gi[HD];
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          qi[c] = bcast(qreg[c], it);
          gi[c] = bcast(greg[c], it);
          s = fmaf(k[c], qi[c], s);
          dp = fmaf(v[c], gi[c], dp);
        }
        const int i = qc * 64 + it;
        const bool ok = vj && (literal || !p.causal || r <= i);
        const float sv = literal ? s / p.sqrt_hd + -10000.0f : s * p.scale;
        const float pj = ok ? __expf(sv - bcast(lq, it)) : 0.f;
        const float mk = attn_keep<DROP>(p, attn_rowkey(p, b, h, i), rr);
        const float ds = pj * (mk * dp - bcast(Dq, it));
        const float pd = pj * mk;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          dk[c] = fmaf(ds, qi[c], dk[c]);
          dv[c] = fmaf(pd, gi[c], dv[c]);
        }
      }
    }
    if (active) {
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        orow[p.d + c] = dk[c] * f;
        orow[2 * p.d + c] = dv[c];
      }
    }
  }
}

long long attn_lse_floats(int B, int H, int L) { return (long long)B * H * L; }
long long attn_bwd_ws_floats(int B, int H, int L) { return (long long)B * H * L + 64; }

// ------------------------------------------------------------------------------------------------------------
// 16x16x4-MFMA attention for ANY sequence length (head dim 4 / 8 / 16), forward.  (Rounds 1-3 also had 32 x 32 single-block kernels
// for L <= 64: they kept the whole [L, L] score block of a (sequence, head) in accumulators and wasted most of every 32-row MFMA on an
// 8-wide head; measured slower at every shape and taken out in round 5.)  The block is walked in 16 x 16 tiles, flash-attention style:
//   * one wave = one (sequence, head, chunk of 64 queries); K and V rows of the keys the chunk can see are staged ONCE in
//     the wave's LDS slice;
//   * S^T tile [16 keys, 16 queries] = K Q^T by HD/4 v_mfma_f32_16x16x4_f32 (lane quarter kq feeds dims kq*HD/4 + s: the k
//     index of an MFMA step is a free permutation); accumulator register r of lane (query c16, kq) is key 4 kq + r, so a lane
//     owns ONE query and four of the tile's keys -- the running max is shared by two shuffles, the running sum stays a
//     per-lane partial (combined once at the end);
//   * O^T [features, queries] += V^T P^T by 4 MFMAs with the probabilities used AS THEY SIT in the accumulators as the B
//     operand (step r <-> keys 4 kq + r); the online-softmax rescale is a per-lane scalar because column = query.
// Masks, dead rows, the literal (-10000) path of an all-padding sequence, compacted rows and dropout are those of the
// kernels above.
typedef float floatx4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int attn_m16_lds_floats_per_wave(int L, int hd) { return 2 * L * (hd + 4) + ((L + 3) & ~3); }

template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_m16_kernel(const float* __restrict__ qkv, const int* __restrict__ seq, AttnDims p,
                                                           float* __restrict__ ctx, float* __restrict__ lse) {
  constexpr int KS = HD / 4;            // MFMA steps of a score tile; also the floats of a row fragment per lane quarter
  constexpr int LDK = HD + 4;           // LDS row stride of the K / V tiles
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  extern __shared__ __attribute__((aligned(16))) float smem_m16[];
  // a workgroup = one (sequence, head): K / V are staged once, the four waves take the 16-query tiles round-robin
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int b, h;
  if (!attn_bh_of_block(p, HD, b, h)) return;
  const int L = p.L, ld = 3 * p.d;
  const int c16 = lane & 15, kq = lane >> 4;
  long long row0;
  int pad;
  seq_rows(p, b, row0, pad);
  const float* __restrict__ base = qkv + row0 * ld + h * HD;
  const int* __restrict__ sq = seq + (long long)b * L;
  const int kend = L, kstage = L;
  float* ks = smem_m16;                                      // [L][LDK]
  float* vs = ks + L * LDK;                                  // [L][LDK]
  float* kvalid = vs + L * LDK;                              // [L] 1 = key may be attended
  // A workgroup lives for a few microseconds, most of them round trips to memory: ids -> first valid key -> K / V rows -> barrier ->
  // query fragment were FOUR dependent trips.  Everything the workgroup reads is requested here, before the first wait: the query
  // fragment of the wave's first tile, the first 256 K / V rows (thread = row), their ids, and the ids first_valid_key looks at.
  float qf0[KS];
  {
    const int irow = max(min(w * 16 + c16, L - 1), pad);
#pragma unroll
    for (int s = 0; s < KS; ++s) qf0[s] = base[(long long)irow * ld + kq * KS + s];
  }
  float4 k0[HD / 4], v0[HD / 4];
  const int j_first = min((int)threadIdx.x, kstage - 1);
  {
    const float* src = base + (long long)max(j_first, pad) * ld;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      k0[c / 4] = *(const float4*)(src + p.d + c);
      v0[c / 4] = *(const float4*)(src + 2 * p.d + c);
    }
  }
  const int sq0 = sq[j_first];
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const bool literal = fv >= L;
  const bool causal = p.causal && !literal;
  // ---- stage K, V rows [0, kstage) and the key mask (thread = row, 16-byte copies)
  if ((int)threadIdx.x < kstage) {
    const int j = threadIdx.x;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      *(float4*)(ks + j * LDK + c) = k0[c / 4];
      *(float4*)(vs + j * LDK + c) = v0[c / 4];
    }
    kvalid[j] = (literal || sq0 > 0) ? 1.f : 0.f;
  }
  for (int j = threadIdx.x + 256; j < kstage; j += 256) {
    const float* src = base + (long long)max(j, pad) * ld;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      *(float4*)(ks + j * LDK + c) = *(const float4*)(src + p.d + c);
      *(float4*)(vs + j * LDK + c) = *(const float4*)(src + 2 * p.d + c);
    }
    kvalid[j] = (literal || sq[j] > 0) ? 1.f : 0.f;
  }
  __syncthreads();
  const float sc2 = p.scale * LOG2E;
  const int jt0 = literal ? 0 : fv >> 4;                     // key tiles before the first valid key hold nothing
  const int nt = (L + 15) >> 4;
  for (int it = w; it < nt; it += 4) {
    const int i = it * 16 + c16;                              // this lane's query
    const int irow = max(min(i, L - 1), pad);
    float qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = qf0[s];
    if (it != w) {
#pragma unroll
      for (int s = 0; s < KS; ++s) qf[s] = base[(long long)irow * ld + kq * KS + s];
    }
    const unsigned rk = attn_rowkey(p, b, h, min(i, L - 1));
    float m = -INFINITY, l = 0.f;
    floatx4 oa = {0.f, 0.f, 0.f, 0.f};
    const int jt_end = causal ? it + 1 : nt;
    for (int jt = jt0; jt < jt_end; ++jt) {
      const int jrow = min(jt * 16 + c16, kend - 1);
      floatx4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) st = __builtin_amdgcn_mfma_f32_16x16x4f32(ks[jrow * LDK + kq * KS + s], qf[s], st, 0, 0, 0);
      float e[4], tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jt * 16 + 4 * kq + r;
        const bool ok = (j < kend) & (kvalid[min(j, kend - 1)] != 0.f) & (!causal | (j <= i));   // (no short-circuit: no divergent branches)
        e[r] = ok ? (literal ? (st[r] / p.sqrt_hd + -10000.0f) * LOG2E : st[r] * sc2) : -INFINITY;
        tmax = fmaxf(tmax, e[r]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m, tmax);
      const float mm = m_new == -INFINITY ? 0.f : m_new;      // nothing visible for this query so far: every exponent is -inf -> 0
      const float corr = __builtin_amdgcn_exp2f(m - mm);      // exp2(-inf) = 0 on the first visible tile (no lane-divergent skip:
      float pr[4];                                            // the MFMAs below take operands from every lane)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = __builtin_amdgcn_exp2f(e[r] - mm);
        l = (r == 0 ? l * corr : l) + pr[r];
        if constexpr (DROP) pr[r] *= attn_keep<DROP>(p, rk, jt * 16 + 4 * kq + r);
      }
      oa[0] *= corr; oa[1] *= corr; oa[2] *= corr; oa[3] *= corr;
      m = m_new;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = min(jt * 16 + 4 * kq + r, kend - 1);
        const float vraw = vs[j * LDK + c16];   // (c16 >= HD reads padding / the next row: discarded by the select)
        oa = __builtin_amdgcn_mfma_f32_16x16x4f32(c16 < HD ? vraw : 0.f, pr[r], oa, 0, 0, 0);
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (i < L && i >= pad) {
      const bool dead = l == 0.f;   // padded-prefix row of a non-empty sequence: unreachable from the loss
      const float inv_l = dead ? 0.f : 1.0f / l;
      if (4 * kq < HD) *(float4*)(ctx + (row0 + i) * p.d + h * HD + 4 * kq) = make_float4(oa[0] * inv_l, oa[1] * inv_l, oa[2] * inv_l, oa[3] * inv_l);
      if (kq == 0) lse[((long long)b * p.H + h) * L + i] = dead ? 0.f : (m + __log2f(l)) * LN2;
    }
  }
}

// 16x16x4-MFMA attention backward for any L (head dim 4 / 8 / 16).  A workgroup = one (sequence, head): Q, K, V, dO rows and the
// per-query lse / D = dO . O are staged once in LDS; the four waves then take the 16-row tiles round-robin, twice:
//   A (lane = query i): S^T, dP^T tiles by MFMA, dS^T = P^T (m dP^T - D_i) lane-locally, dQ^T += K^T dS^T with dS^T as the
//     B operand straight from the accumulators;
//   B (lane = key j):   S, dP tiles (registers run over queries), dV^T += dO^T (P m), dK^T += Q^T dS.
// Nothing [L, L]-shaped is stored; dropout masks are re-evaluated from (row id, column).
__host__ __device__ inline int attn_m16_bwd_lds_floats(int L, int hd) { return 4 * L * (hd + 4) + 3 * ((L + 3) & ~3); }

template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_m16_kernel(const float* __restrict__ qkv, const int* __restrict__ seq,
                                                           const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                           const float* __restrict__ lse, AttnDims p, float* __restrict__ dqkv) {
  constexpr int KS = HD / 4, LDK = HD + 4;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) float smem_m16[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int b, h;
  if (!attn_bh_of_block(p, HD, b, h)) return;
  const int L = p.L, ld = 3 * p.d;
  const int c16 = lane & 15, kq = lane >> 4;
  long long row0;
  int pad;
  seq_rows(p, b, row0, pad);
  const float* __restrict__ base = qkv + row0 * ld + h * HD;
  const int* __restrict__ sq = seq + (long long)b * L;
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const bool literal = fv >= L;
  const bool causal = p.causal && !literal;
  const float f = literal ? 1.0f / p.sqrt_hd : p.scale;   // d(score) / d(q . k)
  const float sc2 = f * LOG2E;
  const int Lp = (L + 3) & ~3;
  float* Qs = smem_m16;
  float* Ks = Qs + L * LDK;
  float* Vs = Ks + L * LDK;
  float* Gs = Vs + L * LDK;
  float* lse2s = Gs + L * LDK;     // lse * log2(e) per query
  float* Ds = lse2s + Lp;          // dO . O per query
  float* kvalid = Ds + Lp;         // 1 = key may be attended
  for (int j = threadIdx.x; j < L; j += 256) {
    const long long row = row0 + max(j, pad);
    const float* src = qkv + row * ld + h * HD;
    const float* gr = dctx + row * p.d + h * HD;
    const float* orr = ctx + row * p.d + h * HD;
    float D = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 g4 = *(const float4*)(gr + c), o4 = *(const float4*)(orr + c);
      *(float4*)(Qs + j * LDK + c) = *(const float4*)(src + c);
      *(float4*)(Ks + j * LDK + c) = *(const float4*)(src + p.d + c);
      *(float4*)(Vs + j * LDK + c) = *(const float4*)(src + 2 * p.d + c);
      *(float4*)(Gs + j * LDK + c) = g4;
      D += (g4.x * o4.x + g4.y * o4.y) + (g4.z * o4.z + g4.w * o4.w);
    }
    lse2s[j] = lse[((long long)b * p.H + h) * L + j] * LOG2E;
    Ds[j] = D;
    kvalid[j] = (literal || sq[j] > 0) ? 1.f : 0.f;
  }
  __syncthreads();
  (void)base;
  const int nt = (L + 15) >> 4;
  const int jt0 = literal ? 0 : fv >> 4;
  float* orow = dqkv + row0 * ld + h * HD;
  // =========================================================================== phase A: lane = query
  for (int it = w; it < nt; it += 4) {
    const int i = it * 16 + c16, ic = min(i, L - 1);
    float qf[KS], gf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { qf[s] = Qs[ic * LDK + kq * KS + s]; gf[s] = Gs[ic * LDK + kq * KS + s]; }
    const float lse2 = lse2s[ic], Di = Ds[ic];
    const unsigned rk = attn_rowkey(p, b, h, ic);
    floatx4 dq = {0.f, 0.f, 0.f, 0.f};
    const int jt_end = causal ? it + 1 : nt;
    for (int jt = jt0; jt < jt_end; ++jt) {
      const int jr = min(jt * 16 + c16, L - 1);
      floatx4 sT = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sT = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[jr * LDK + kq * KS + s], qf[s], sT, 0, 0, 0);
        dpT = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[jr * LDK + kq * KS + s], gf[s], dpT, 0, 0, 0);
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jt * 16 + 4 * kq + r, jc = min(j, L - 1);
        const bool ok = (j < L) & (kvalid[jc] != 0.f) & (!causal | (j <= i));   // (no short-circuit: no divergent branches)
        const float e = literal ? (sT[r] / p.sqrt_hd + -10000.0f) * LOG2E : sT[r] * sc2;
        const float pv = __builtin_amdgcn_exp2f(ok ? e - lse2 : -INFINITY);   // branch-free: exp2(-inf) = 0 for masked keys
        ds[r] = pv * (attn_keep<DROP>(p, rk, j) * dpT[r] - Di);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jc = min(jt * 16 + 4 * kq + r, L - 1);
        const float kk = Ks[jc * LDK + c16];   // (c16 >= HD reads the row's padding / the next row: discarded by the select)
        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(c16 < HD ? kk : 0.f, ds[r], dq, 0, 0, 0);
      }
    }
    if (i < L && i >= pad && 4 * kq < HD)
      *(float4*)(orow + (long long)i * ld + 4 * kq) = make_float4(dq[0] * f, dq[1] * f, dq[2] * f, dq[3] * f);
  }
  // =========================================================================== phase B: lane = key
  for (int jt = w; jt < nt; jt += 4) {   // key tile jt meets nt - jt query tiles, query tile it met it + 1 key tiles: same w balances the two phases
    const int j = jt * 16 + c16, jc = min(j, L - 1);
    float kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { kf[s] = Ks[jc * LDK + kq * KS + s]; vf[s] = Vs[jc * LDK + kq * KS + s]; }
    const bool kv = (j < L) & (kvalid[jc] != 0.f);
    floatx4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    for (int it = causal ? jt : 0; it < nt; ++it) {
      const int ir = min(it * 16 + c16, L - 1);
      floatx4 sM = {0.f, 0.f, 0.f, 0.f}, dpM = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sM = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[ir * LDK + kq * KS + s], kf[s], sM, 0, 0, 0);
        dpM = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[ir * LDK + kq * KS + s], vf[s], dpM, 0, 0, 0);
      }
      float pd[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = it * 16 + 4 * kq + r, ic = min(i, L - 1);
        const bool ok = kv & (i < L) & (i >= pad) & (!causal | (j <= i));
        const float e = literal ? (sM[r] / p.sqrt_hd + -10000.0f) * LOG2E : sM[r] * sc2;
        const float pv = __builtin_amdgcn_exp2f(ok ? e - lse2s[ic] : -INFINITY);
        const float mk = DROP ? drop_mul(attn_rowkey(p, b, h, ic), (unsigned)j, p.dthresh, p.dscale) : 1.0f;
        pd[r] = pv * mk;
        ds[r] = pv * (mk * dpM[r] - Ds[ic]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ic = min(it * 16 + 4 * kq + r, L - 1);
        const float gg = Gs[ic * LDK + c16], qq = Qs[ic * LDK + c16];
        dv = __builtin_amdgcn_mfma_f32_16x16x4f32(c16 < HD ? gg : 0.f, pd[r], dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_16x16x4f32(c16 < HD ? qq : 0.f, ds[r], dk, 0, 0, 0);
      }
    }
    if (j < L && j >= pad && 4 * kq < HD) {
      float* out = orow + (long long)j * ld;
      *(float4*)(out + p.d + 4 * kq) = make_float4(dk[0] * f, dk[1] * f, dk[2] * f, dk[3] * f);
      *(float4*)(out + 2 * p.d + 4 * kq) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    }
  }
}

// Short-sequence variant of the backward above (same tiling, same arithmetic per element; used while its LDS footprint leaves
// >= 4 workgroups per CU).  The backward at L = 50 is VALU-issue bound, not MFMA or HBM bound, so this one spends LDS to shed
// instructions: the staged rows are padded to whole 16-row tiles (zero rows, validity 0 -> no index clamps and no j < L tests),
// the per-row scalars (key / query validity, lse, D) are read four at a time, and K, Q, dO are also kept TRANSPOSED and
// zero-padded to 16 feature rows, so the A operands of the dQ / dK / dV MFMAs are one ds_read_b128 instead of four reads and
// four selects.  The all-padding ("literal") path is a compile-time copy of the tile loops instead of a branch per element.
__host__ __device__ inline int attn_m16t_lds_floats(int L, int hd) {
  const int Lp = (L + 15) & ~15;
  return 4 * Lp * (hd + 4) + 3 * 16 * (Lp + 4) + 4 * Lp;
}

template <int KS>
__device__ __forceinline__ void lds_frag(const float* __restrict__ src, float (&out)[KS]) {
  if constexpr (KS == 1) out[0] = src[0];
  else if constexpr (KS == 2) { const float2 v = *(const float2*)src; out[0] = v.x; out[1] = v.y; }
  else { const float4 v = *(const float4*)src; out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
}

template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_m16t_kernel(const float* __restrict__ qkv, const int* __restrict__ seq,
                                                            const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                            const float* __restrict__ lse, AttnDims p, float* __restrict__ dqkv) {
  constexpr int KS = HD / 4, LDK = HD + 4, PR = HD / 4;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) float smem_m16[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int b, h;
  if (!attn_bh_of_block(p, HD, b, h)) return;
  const int L = p.L, ld = 3 * p.d;
  const int c16 = lane & 15, kq = lane >> 4;
  long long row0;
  int pad;
  seq_rows(p, b, row0, pad);
  const int* __restrict__ sq = seq + (long long)b * L;
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const bool literal = fv >= L;
  const bool causal = p.causal && !literal;
  const float f = literal ? 1.0f / p.sqrt_hd : p.scale;   // d(score) / d(q . k)
  const int nt = (L + 15) >> 4, Lp = nt * 16, S = Lp + 4;
  float* Qs = smem_m16;              // [Lp][LDK] rows (rows >= L zero)
  float* Ks = Qs + Lp * LDK;
  float* Vs = Ks + Lp * LDK;
  float* Gs = Vs + Lp * LDK;
  float* KT = Gs + Lp * LDK;         // [16][S] transposed (feature rows >= HD and columns >= L zero)
  float* QT = KT + 16 * S;
  float* GT = QT + 16 * S;
  float* lse2s = GT + 16 * S;        // [Lp] lse * log2(e) per query
  float* Ds = lse2s + Lp;            // [Lp] dO . O per query
  float* kvalid = Ds + Lp;           // [Lp] 1 = key may be attended
  float* qvalid = kvalid + Lp;       // [Lp] 1 = query row exists (pad <= i < L)
  for (int x = threadIdx.x; x < Lp * PR; x += 256) {
    const int j = x / PR, c = (x % PR) * 4;
    const bool in = j < L;
    const long long row = row0 + max(min(j, L - 1), pad);
    const float* src = qkv + row * ld + h * HD + c;
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, v4 = q4, g4 = q4, o4 = q4;
    if (in) {
      q4 = *(const float4*)src; k4 = *(const float4*)(src + p.d); v4 = *(const float4*)(src + 2 * p.d);
      g4 = *(const float4*)(dctx + row * p.d + h * HD + c); o4 = *(const float4*)(ctx + row * p.d + h * HD + c);
    }
    *(float4*)(Qs + j * LDK + c) = q4;
    *(float4*)(Ks + j * LDK + c) = k4;
    *(float4*)(Vs + j * LDK + c) = v4;
    *(float4*)(Gs + j * LDK + c) = g4;
    float* kt = KT + c * S + j; float* qt = QT + c * S + j; float* gt = GT + c * S + j;
    kt[0] = k4.x; kt[S] = k4.y; kt[2 * S] = k4.z; kt[3 * S] = k4.w;
    qt[0] = q4.x; qt[S] = q4.y; qt[2 * S] = q4.z; qt[3 * S] = q4.w;
    gt[0] = g4.x; gt[S] = g4.y; gt[2 * S] = g4.z; gt[3 * S] = g4.w;
    float D = (g4.x * o4.x + g4.y * o4.y) + (g4.z * o4.z + g4.w * o4.w);
#pragma unroll
    for (int o = 1; o < PR; o <<= 1) D += __shfl_xor(D, o, 64);   // the PR lanes of a row are adjacent and active together
    if ((x % PR) == 0) {
      lse2s[j] = in ? lse[((long long)b * p.H + h) * L + j] * LOG2E : 0.f;
      Ds[j] = D;
      kvalid[j] = (in && (literal || sq[j] > 0)) ? 1.f : 0.f;
      qvalid[j] = (in && j >= pad) ? 1.f : 0.f;
    }
  }
  if constexpr (HD < 16) {   // feature rows HD..15 of the transposed copies
    const int n4 = (16 - HD) * S / 4;
    for (int x = threadIdx.x; x < 3 * n4; x += 256) {
      float* dst = (x < n4 ? KT : x < 2 * n4 ? QT : GT) + HD * S + (x % n4) * 4;
      *(float4*)dst = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  const int jt0 = literal ? 0 : fv >> 4;
  float* orow = dqkv + row0 * ld + h * HD;
  const float sc2 = f * LOG2E, inv_sqrt_div = p.sqrt_hd;
  auto tiles = [&](auto lit_tag) {
    constexpr bool LIT = decltype(lit_tag)::value;
    auto expo = [&](float sv) { return LIT ? (sv / inv_sqrt_div + -10000.0f) * LOG2E : sv * sc2; };
    // ========================================================================= phase A: lane = query
    for (int it = w; it < nt; it += 4) {
      const int i = it * 16 + c16;
      float qf[KS], gf[KS];
      lds_frag<KS>(Qs + i * LDK + kq * KS, qf);
      lds_frag<KS>(Gs + i * LDK + kq * KS, gf);
      const float lse2 = lse2s[i], Di = Ds[i];
      const unsigned rk = attn_rowkey(p, b, h, i);
      floatx4 dq = {0.f, 0.f, 0.f, 0.f};
      const int jt_end = (!LIT && causal) ? it + 1 : nt;
      for (int jt = jt0; jt < jt_end; ++jt) {
        float kf[KS], vf[KS];
        lds_frag<KS>(Ks + (jt * 16 + c16) * LDK + kq * KS, kf);
        lds_frag<KS>(Vs + (jt * 16 + c16) * LDK + kq * KS, vf);
        const int j0 = jt * 16 + 4 * kq;
        const float4 kv4 = *(const float4*)(kvalid + j0);
        const float4 kt4 = *(const float4*)(KT + c16 * S + j0);
        floatx4 sT = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          sT = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], sT, 0, 0, 0);
          dpT = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s], gf[s], dpT, 0, 0, 0);
        }
        const float kvr[4] = {kv4.x, kv4.y, kv4.z, kv4.w}, ktr[4] = {kt4.x, kt4.y, kt4.z, kt4.w};
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + r;
          const bool ok = (kvr[r] != 0.f) & ((LIT || !causal) | (j <= i));   // (no short-circuit: no divergent branches)
          const float pv = __builtin_amdgcn_exp2f(ok ? expo(sT[r]) - lse2 : -INFINITY);   // exp2(-inf) = 0 for masked keys
          ds[r] = pv * (attn_keep<DROP>(p, rk, j) * dpT[r] - Di);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dq = __builtin_amdgcn_mfma_f32_16x16x4f32(ktr[r], ds[r], dq, 0, 0, 0);
      }
      if (i < L && i >= pad && 4 * kq < HD)
        *(float4*)(orow + (long long)i * ld + 4 * kq) = make_float4(dq[0] * f, dq[1] * f, dq[2] * f, dq[3] * f);
    }
    // ========================================================================= phase B: lane = key
    for (int jt = w; jt < nt; jt += 4) {   // key tile jt meets nt - jt query tiles, query tile it met it + 1 key tiles
      const int j = jt * 16 + c16;
      float kf[KS], vf[KS];
      lds_frag<KS>(Ks + j * LDK + kq * KS, kf);
      lds_frag<KS>(Vs + j * LDK + kq * KS, vf);
      const bool kv = kvalid[j] != 0.f;
      floatx4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
      for (int it = (!LIT && causal) ? jt : 0; it < nt; ++it) {
        float qf[KS], gf[KS];
        lds_frag<KS>(Qs + (it * 16 + c16) * LDK + kq * KS, qf);
        lds_frag<KS>(Gs + (it * 16 + c16) * LDK + kq * KS, gf);
        const int i0 = it * 16 + 4 * kq;
        const float4 qv4 = *(const float4*)(qvalid + i0), l4 = *(const float4*)(lse2s + i0), d4 = *(const float4*)(Ds + i0);
        const float4 gt4 = *(const float4*)(GT + c16 * S + i0), qt4 = *(const float4*)(QT + c16 * S + i0);
        floatx4 sM = {0.f, 0.f, 0.f, 0.f}, dpM = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          sM = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], kf[s], sM, 0, 0, 0);
          dpM = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[s], vf[s], dpM, 0, 0, 0);
        }
        const float qvr[4] = {qv4.x, qv4.y, qv4.z, qv4.w}, lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
        const float gtr[4] = {gt4.x, gt4.y, gt4.z, gt4.w}, qtr[4] = {qt4.x, qt4.y, qt4.z, qt4.w};
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + r;
          const bool ok = kv & (qvr[r] != 0.f) & ((LIT || !causal) | (j <= i));
          const float pv = __builtin_amdgcn_exp2f(ok ? expo(sM[r]) - lr[r] : -INFINITY);
          const float mk = DROP ? drop_mul(attn_rowkey(p, b, h, i), (unsigned)j, p.dthresh, p.dscale) : 1.0f;
          pd[r] = pv * mk;
          ds[r] = pv * (mk * dpM[r] - dr[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dv = __builtin_amdgcn_mfma_f32_16x16x4f32(gtr[r], pd[r], dv, 0, 0, 0);
          dk = __builtin_amdgcn_mfma_f32_16x16x4f32(qtr[r], ds[r], dk, 0, 0, 0);
        }
      }
      if (j < L && j >= pad && 4 * kq < HD) {
        float* out = orow + (long long)j * ld;
        *(float4*)(out + p.d + 4 * kq) = make_float4(dk[0] * f, dk[1] * f, dk[2] * f, dk[3] * f);
        *(float4*)(out + 2 * p.d + 4 * kq) = make_float4(dv[0], dv[1], dv[2], dv[3]);
      }
    }
  };
  if (literal) tiles(std::true_type{}); else tiles(std::false_type{});
}

// ---- L <= 64, a WAVE per head: four heads per workgroup, every (query tile, key tile) pair evaluated once.
// What the kernel above pays for, measured at C5 (B = 512, L = 50, 16 heads of 8; 63 us): 31 us are the STAGING alone -- a head's
// rows of Q, K, V, dO, O are 32-byte pieces of 128-byte lines, so each workgroup pulls whole lines into its L1 for a quarter of
// their bytes (262 MB through the L1s for 67 MB of operands: the XCD placement makes the four heads of a line share an L2, not an
// L1) -- 13 us the query-oriented pass, 19 us the key-oriented pass, which re-evaluates every score, exp2, mask and dP.  Here
//   * a workgroup stages FOUR adjacent heads: 8 threads read one row's 128-byte line of each tensor, nothing is fetched twice;
//   * wave w owns head h0 + w entirely: all its tile pairs (balanced: every wave has the same pairs), dQ, dK and dV accumulate in
//     ITS registers (dK / dV: one accumulator pair per key tile, <= 4 tiles) -- no cross-wave reduction, one barrier in the kernel;
//   * a pair is evaluated ONCE, in the query orientation (lane = query): P and dS, 16 x 16 as they sit in the accumulators, go through
//     a per-wave LDS scratch (2 x 16 x 20 floats: one ds_write_b128 each, four conflict-free ds_read_b32 back) and are then the
//     B operands of the dK / dV MFMAs in the key orientation.  2 KS + 12 MFMAs per pair instead of 4 KS + 12; 4 exp2 / mask /
//     dropout evaluations per lane instead of 8;
//   * no transposed copies of K, Q, dO: the A operands of the dQ / dK / dV MFMAs are read from the row-major tiles (conflict-free:
//     rows 48 bytes apart) with a select for the feature rows >= HD.
// Same arithmetic per element as the kernels above; results differ from theirs only by the summation order inside dK / dV.
constexpr int M16W_TS = 20, M16W_HEADS = 4;
// row stride of the staged tiles: head dim 8 needs no padding -- rows 8 .. 15 of every 16 keep their two 16-byte halves swapped
// (column c of row r sits at r * 8 + (c ^ ((r & 8) >> 1))), which makes the b64 fragment reads of a 16-row tile conflict-free; the
// 8 floats per row this saves (12 -> 8) are what lets TWO of these workgroups share a CU with a weight-gradient GEMM of the side
// stream (2 x 45 + 67 KB <= 160 KB) instead of one
__host__ __device__ inline int attn_m16w_ldk(int hd) { return hd == 8 ? 8 : hd + 4; }
__host__ __device__ inline int attn_m16w_head_floats(int L, int hd) {
  const int Lp = (L + 15) & ~15;
  return 4 * Lp * attn_m16w_ldk(hd) + 2 * Lp;
}
__host__ __device__ inline int attn_m16w_lds_floats(int L, int hd) {
  const int Lp = (L + 15) & ~15;
  return M16W_HEADS * attn_m16w_head_floats(L, hd) + 2 * Lp + M16W_HEADS * 2 * 16 * M16W_TS;
}

template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_m16w_kernel(const float* __restrict__ qkv, const int* __restrict__ seq,
                                                            const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                            const float* __restrict__ lse, AttnDims p, float* __restrict__ dqkv) {
  constexpr int KS = HD / 4, LDK = HD == 8 ? 8 : HD + 4, PR = HD / 4, NTM = 4, TS = M16W_TS, SL = M16W_HEADS * PR;   // SL: float4 slots per row
  auto at = [](int row, int c) { return row * LDK + (HD == 8 ? (c ^ ((row & 8) >> 1)) : c); };   // element (row, c) of a staged tile
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) float smem_m16[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int hg = (p.H + M16W_HEADS - 1) / M16W_HEADS;
  const int b = blockIdx.x / hg, h0 = (blockIdx.x % hg) * M16W_HEADS;
  const int L = p.L, ld = 3 * p.d;
  const int c16 = lane & 15, kq = lane >> 4;
  long long row0;
  int pad;
  seq_rows(p, b, row0, pad);
  const int* __restrict__ sq = seq + (long long)b * L;
  const int fv = UR_UNIFORM(first_valid_key(sq, L, lane));
  const bool literal = fv >= L;
  const bool causal = p.causal && !literal;
  const float f = literal ? 1.0f / p.sqrt_hd : p.scale;   // d(score) / d(q . k)
  const int nt = (L + 15) >> 4, Lp = nt * 16;
  const int HF = 4 * Lp * LDK + 2 * Lp;              // floats per head: Q, K, V, dO tiles [Lp][LDK] (rows >= L zero), lse2, D
  float* kvalid = smem_m16 + M16W_HEADS * HF;        // [Lp] 1 = key may be attended
  float* qvalid = kvalid + Lp;                       // [Lp] 1 = query row exists (pad <= i < L)
  float* TP = qvalid + Lp + w * (2 * 16 * TS);       // this wave's transpose scratch: P tile, dS tile
  float* TD = TP + 16 * TS;
  for (int x = threadIdx.x; x < Lp * SL; x += 256) {
    const int j = x / SL, q = x % SL, hh = q / PR, c = (q % PR) * 4, h = h0 + hh;
    const bool in = j < L && h < p.H;
    const long long row = row0 + max(min(j, L - 1), pad);
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, v4 = q4, g4 = q4, o4 = q4;
    if (in) {
      const float* src = qkv + row * ld + h * HD + c;
      q4 = *(const float4*)src; k4 = *(const float4*)(src + p.d); v4 = *(const float4*)(src + 2 * p.d);
      g4 = *(const float4*)(dctx + row * p.d + h * HD + c); o4 = *(const float4*)(ctx + row * p.d + h * HD + c);
    }
    float* base = smem_m16 + hh * HF;
    *(float4*)(base + at(j, c)) = q4;
    *(float4*)(base + Lp * LDK + at(j, c)) = k4;
    *(float4*)(base + 2 * Lp * LDK + at(j, c)) = v4;
    *(float4*)(base + 3 * Lp * LDK + at(j, c)) = g4;
    float D = (g4.x * o4.x + g4.y * o4.y) + (g4.z * o4.z + g4.w * o4.w);
#pragma unroll
    for (int o = 1; o < PR; o <<= 1) D += __shfl_xor(D, o, 64);   // the PR lanes of a (row, head) are adjacent and active together
    if ((q % PR) == 0) {
      base[4 * Lp * LDK + j] = in ? lse[((long long)b * p.H + h) * L + j] * LOG2E : 0.f;
      base[4 * Lp * LDK + Lp + j] = D;
    }
    if (q == 0) {
      kvalid[j] = (j < L && (literal || sq[j] > 0)) ? 1.f : 0.f;
      qvalid[j] = (j < L && j >= pad) ? 1.f : 0.f;
    }
  }
  __syncthreads();
  const int h = h0 + w;
  if (h >= p.H) return;
  const float* Qs = smem_m16 + w * HF;
  const float* Ks = Qs + Lp * LDK;
  const float* Vs = Ks + Lp * LDK;
  const float* Gs = Vs + Lp * LDK;
  const float* lse2s = Gs + Lp * LDK;
  const float* Ds = lse2s + Lp;
  const int jt0 = literal ? 0 : fv >> 4;
  float* orow = dqkv + row0 * ld + h * HD;
  const float sc2 = f * LOG2E, inv_sqrt_div = p.sqrt_hd;
  const bool frow = c16 < HD;                       // this lane's feature row of an A operand exists
  const int cA = frow ? c16 : 0;
  floatx4 dkA[NTM], dvA[NTM];
#pragma unroll
  for (int q = 0; q < NTM; ++q) {
    dkA[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    dvA[q] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  auto tiles = [&](auto lit_tag) {
    constexpr bool LIT = decltype(lit_tag)::value;
    auto expo = [&](float sv) { return LIT ? (sv / inv_sqrt_div + -10000.0f) * LOG2E : sv * sc2; };
    for (int it = 0; it < nt; ++it) {
      const int i = it * 16 + c16;
      float qf[KS], gf[KS];
      lds_frag<KS>(Qs + at(i, kq * KS), qf);
      lds_frag<KS>(Gs + at(i, kq * KS), gf);
      const float lse2 = lse2s[i], Di = Ds[i];
      const bool qv = qvalid[i] != 0.f;
      const unsigned rk = attn_rowkey(p, b, h, i);
      const int i0 = it * 16 + 4 * kq;
      float gtr[4], qtr[4];                          // A operands of the dV / dK MFMAs: dO / Q feature c16 of queries i0 .. i0 + 3
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gg = Gs[at(i0 + r, cA)], qq = Qs[at(i0 + r, cA)];
        gtr[r] = frow ? gg : 0.f;
        qtr[r] = frow ? qq : 0.f;
      }
      floatx4 dq = {0.f, 0.f, 0.f, 0.f};
      const int jt_end = (!LIT && causal) ? it + 1 : nt;
#pragma unroll
      for (int jt = 0; jt < NTM; ++jt) {
        if (jt < jt0 || jt >= jt_end) continue;     // (wave-uniform)
        float kf[KS], vf[KS];
        lds_frag<KS>(Ks + at(jt * 16 + c16, kq * KS), kf);
        lds_frag<KS>(Vs + at(jt * 16 + c16, kq * KS), vf);
        const int j0 = jt * 16 + 4 * kq;
        const float4 kv4 = *(const float4*)(kvalid + j0);
        float ktr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float kk = Ks[at(j0 + r, cA)];
          ktr[r] = frow ? kk : 0.f;
        }
        floatx4 sT = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
          sT = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s2], qf[s2], sT, 0, 0, 0);
          dpT = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s2], gf[s2], dpT, 0, 0, 0);
        }
        const float kvr[4] = {kv4.x, kv4.y, kv4.z, kv4.w};
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + r;
          const bool ok = qv & (kvr[r] != 0.f) & ((LIT || !causal) | (j <= i));   // (no short-circuit: no divergent branches)
          const float pv = __builtin_amdgcn_exp2f(ok ? expo(sT[r]) - lse2 : -INFINITY);   // exp2(-inf) = 0 for masked pairs
          const float mk = attn_keep<DROP>(p, rk, j);
          pd[r] = pv * mk;
          ds[r] = pv * (mk * dpT[r] - Di);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dq = __builtin_amdgcn_mfma_f32_16x16x4f32(ktr[r], ds[r], dq, 0, 0, 0);
        // query orientation -> key orientation: lane (query c16, keys 4 kq + r) writes a row chunk, lane (key c16, queries 4 kq + r) reads a column
        *(float4*)(TP + c16 * TS + 4 * kq) = make_float4(pd[0], pd[1], pd[2], pd[3]);
        *(float4*)(TD + c16 * TS + 4 * kq) = make_float4(ds[0], ds[1], ds[2], ds[3]);
        __builtin_amdgcn_wave_barrier();
        float pk[4], dsk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pk[r] = TP[(4 * kq + r) * TS + c16];
          dsk[r] = TD[(4 * kq + r) * TS + c16];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dvA[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gtr[r], pk[r], dvA[jt], 0, 0, 0);
          dkA[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qtr[r], dsk[r], dkA[jt], 0, 0, 0);
        }
      }
      if (i < L && i >= pad && 4 * kq < HD)
        *(float4*)(orow + (long long)i * ld + 4 * kq) = make_float4(dq[0] * f, dq[1] * f, dq[2] * f, dq[3] * f);
    }
  };
  if (literal) tiles(std::true_type{}); else tiles(std::false_type{});
  if (4 * kq < HD) {
#pragma unroll
    for (int jt = 0; jt < NTM; ++jt) {
      const int j = jt * 16 + c16;
      if (jt < nt && j < L && j >= pad) {
        float* out = orow + (long long)j * ld;
        *(float4*)(out + p.d + 4 * kq) = make_float4(dkA[jt][0] * f, dkA[jt][1] * f, dkA[jt][2] * f, dkA[jt][3] * f);
        *(float4*)(out + 2 * p.d + 4 * kq) = make_float4(dvA[jt][0], dvA[jt][1], dvA[jt][2], dvA[jt][3]);
      }
    }
  }
}

// launches KERNEL<HD, true> when dropout is on (p.dthresh != 0), KERNEL<HD, false> otherwise
#define UR_ATTN_LAUNCH(KERNEL, HD, ...)                                   \
  do {                                                                    \
    if (p.dthresh) UR_LAUNCH_EV((KERNEL<HD, true>), __VA_ARGS__);         \
    else UR_LAUNCH_EV((KERNEL<HD, false>), __VA_ARGS__);                  \
  } while (0)

static int make_dims(int B, int L, int d, int H, int causal, AttnDims* p) {
  if (H <= 0 || d % H) return fail(UR_ERR_ARG, "attention: d=%d not divisible by n_heads=%d", d, H);
  const int hd = d / H;
  if (hd > 64 || (hd & (hd - 1)) || hd < 2)
    return fail(UR_ERR_UNSUPPORTED, "attention: head dim %d (= d/n_heads) must be a power of two in [2, 64]", hd);
  p->B = B; p->L = L; p->d = d; p->H = H; p->hd = hd; p->causal = causal;
  p->nchunk = (L + 63) / 64;
  p->sqrt_hd = sqrtf((float)hd);
  p->scale = 1.0f / p->sqrt_hd;
  p->seq_base = nullptr;
  p->seq_pad = nullptr;
  p->dkey = 0; p->dthresh = 0; p->dscale = 1.0f;
  return UR_OK;
}

// test hooks (common.h: UR_TEST): attn_no_mfma = the VALU kernels at every shape (what head dims 2 / 32 / 64 and sequences beyond the LDS
// budget take anyway; no compact rows); attn_no_m16 = the same through the 16x16 kernels' own gate
static bool attn_no_mfma() {
  static const bool off = ur_test_hook("attn_no_mfma") != 0;
  return off;
}
static bool attn_m16_supported(int L, int hd) {
  static const bool off = ur_test_hook("attn_no_m16") != 0;
  return !off && (hd == 4 || hd == 8 || hd == 16) && (long long)attn_m16_lds_floats_per_wave(L, hd) * 4 <= 64 * 1024;
}
bool attn_compact_supported(int L, int d, int H) {
  if (H <= 0 || d % H) return false;
  const int hd = d / H;
  if (attn_no_mfma() || !(hd == 4 || hd == 8 || hd == 16)) return false;
  return attn_m16_supported(L, hd) && (long long)attn_m16_bwd_lds_floats(L, hd) * 4 <= 80 * 1024;   // 16x16-tile kernels, both passes
}

int attn_fwd(const float* qkv, const int* seq, int B, int L, int d, int H, int causal, float* ctx, float* lse,
             int q_last_only, hipStream_t st, const int* seq_base, const int* seq_pad, const DropSpec* drop) {
  if (q_last_only) return fail(UR_ERR_UNSUPPORTED, "attn_fwd: last-row mode not implemented");
  ProfScope ps(PC_ATTN_FWD, st, 4.0 * B * L * (double)L * d * (causal ? 0.5 : 1.0));
  AttnDims p;
  int rc = make_dims(B, L, d, H, causal, &p);
  if (rc) return rc;
  p.seq_base = seq_base;
  p.seq_pad = seq_pad;
  if (drop && drop->thresh) { p.dkey = drop->key; p.dthresh = drop->thresh; p.dscale = drop->scale; }
  if (seq_base && !attn_compact_supported(L, d, H)) return fail(UR_ERR_UNSUPPORTED, "attention: compacted rows need head dim 4/8/16 and a sequence that fits the MFMA kernels");
  if (attn_m16_supported(L, p.hd) && !attn_no_mfma()) {
    const size_t lds = (size_t)attn_m16_lds_floats_per_wave(L, p.hd) * sizeof(float);
    dim3 g3(attn_bh_grid(B, H, p.hd));
#define GM(HD)                                                                                                                     \
    do {                                                                                                                             \
      static const hipError_t a0 = hipFuncSetAttribute((const void*)attn_fwd_m16_kernel<HD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      static const hipError_t a1 = hipFuncSetAttribute((const void*)attn_fwd_m16_kernel<HD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
      (void)a0; (void)a1;                                                                                                            \
      UR_ATTN_LAUNCH(attn_fwd_m16_kernel, HD, g3, dim3(256), lds, st, qkv, seq, p, ctx, lse);                                        \
    } while (0)
    if (p.hd == 4) GM(4); else if (p.hd == 8) GM(8); else GM(16);
#undef GM
    UR_LAUNCH_CHECK();
    return UR_OK;
  }
  dim3 grid(B, cdiv(H * p.nchunk, 4));
#define GO(HD) UR_ATTN_LAUNCH(attn_fwd_kernel, HD, grid, dim3(256), 0, st, qkv, seq, p, ctx, lse)
#define GR(HD) UR_ATTN_LAUNCH(attn_fwd_rl_kernel, HD, grid, dim3(256), 0, st, qkv, seq, p, ctx, lse)
  switch (p.hd) {   // small heads: register-broadcast kernels; large heads: scalar-load kernels
    case 2: GR(2); break;
    case 4: GR(4); break;
    case 8: GR(8); break;
    case 16: GR(16); break;
    case 32: GO(32); break;
    default: GO(64); break;
  }
#undef GO
#undef GR
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int attn_bwd(const float* qkv, const int* seq, const float* ctx, const float* dctx, const float* lse, int B, int L, int d,
             int H, int causal, float* dqkv, float* ws, int q_last_only, hipStream_t st, const int* seq_base, const int* seq_pad,
             const DropSpec* drop) {
  if (q_last_only) return fail(UR_ERR_UNSUPPORTED, "attn_bwd: last-row mode not implemented");
  ProfScope ps(PC_ATTN_BWD, st, 10.0 * B * L * (double)L * d * (causal ? 0.5 : 1.0));
  AttnDims p;
  int rc = make_dims(B, L, d, H, causal, &p);
  if (rc) return rc;
  p.seq_base = seq_base;
  p.seq_pad = seq_pad;
  if (drop && drop->thresh) { p.dkey = drop->key; p.dthresh = drop->thresh; p.dscale = drop->scale; }
  if (seq_base && !attn_compact_supported(L, d, H)) return fail(UR_ERR_UNSUPPORTED, "attention: compacted rows need head dim 4/8/16 and a sequence that fits the MFMA kernels");
  if (attn_compact_supported(L, d, H)) {   // (= the 16x16-tile kernels take the shape, both passes)
    dim3 g3(attn_bh_grid(B, H, p.hd));
    static const bool no_t = ur_test_hook("attn_no_m16t") != 0;   // test hook: the un-transposed kernel for short L too
    const size_t lds_t = (size_t)attn_m16t_lds_floats(L, p.hd) * sizeof(float);
    static const bool no_w = ur_test_hook("attn_no_m16w") != 0;   // test hook: the workgroup-per-head kernels for L <= 64 as well
    const size_t lds_w = (size_t)attn_m16w_lds_floats(L, p.hd) * sizeof(float);
    if (!no_t && !no_w && L <= 64 && lds_w <= 64 * 1024) {   // four heads per workgroup, a wave per head, one pass over the tile pairs
      dim3 gw(B * cdiv(H, M16W_HEADS));
#define GW_(HD) UR_ATTN_LAUNCH(attn_bwd_m16w_kernel, HD, gw, dim3(256), lds_w, st, qkv, seq, ctx, dctx, lse, p, dqkv)
      if (p.hd == 4) GW_(4); else if (p.hd == 8) GW_(8); else GW_(16);
#undef GW_
      UR_LAUNCH_CHECK();
      return UR_OK;
    }
    if (!no_t && lds_t <= 40 * 1024) {   // >= 4 workgroups per CU
#define GT_(HD) UR_ATTN_LAUNCH(attn_bwd_m16t_kernel, HD, g3, dim3(256), lds_t, st, qkv, seq, ctx, dctx, lse, p, dqkv)
      if (p.hd == 4) GT_(4); else if (p.hd == 8) GT_(8); else GT_(16);
#undef GT_
      UR_LAUNCH_CHECK();
      return UR_OK;
    }
    const size_t lds = (size_t)attn_m16_bwd_lds_floats(L, p.hd) * sizeof(float);
#define GM(HD)                                                                                                                     \
    do {                                                                                                                             \
      static const hipError_t a0 = hipFuncSetAttribute((const void*)attn_bwd_m16_kernel<HD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      static const hipError_t a1 = hipFuncSetAttribute((const void*)attn_bwd_m16_kernel<HD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
      (void)a0; (void)a1;                                                                                                            \
      UR_ATTN_LAUNCH(attn_bwd_m16_kernel, HD, g3, dim3(256), lds, st, qkv, seq, ctx, dctx, lse, p, dqkv);                             \
    } while (0)
    if (p.hd == 4) GM(4); else if (p.hd == 8) GM(8); else GM(16);
#undef GM
    UR_LAUNCH_CHECK();
    return UR_OK;
  }
  float* Dd = ws;
  dim3 grid(B, cdiv(H * p.nchunk, 4));
  if (p.hd > 16) {
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(B), dim3(256), 0, st, ctx, dctx, p, Dd);
    UR_LAUNCH_CHECK();
  }
#define GO(HD) UR_ATTN_LAUNCH(attn_bwd_kernel, HD, grid, dim3(256), 0, st, qkv, seq, dctx, lse, Dd, p, dqkv)
#define GR(HD) UR_ATTN_LAUNCH(attn_bwd_rl_kernel, HD, grid, dim3(256), 0, st, qkv, seq, ctx, dctx, lse, p, dqkv)
  switch (p.hd) {
    case 2: GR(2); break;
    case 4: GR(4); break;
    case 8: GR(8); break;
    case 16: GR(16); break;
    case 32: GO(32); break;
    default: GO(64); break;
  }
#undef GO
#undef GR
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int attn_last_fwd(const float* q_last, const float* qkv, const int* seq, int B, int L, int d, int H, float* ctx_last,
                  float* lse_last, hipStream_t st, const int* seq_base, const int* seq_pad, const DropSpec* drop, const AttnQProj* qp) {
  const AttnQProj qpv = qp ? *qp : AttnQProj{};
  ProfScope ps(PC_ATTN_FWD, st, 4.0 * B * (double)L * d);
  AttnDims p;
  int rc = make_dims(B, L, d, H, 1, &p);
  if (rc) return rc;
  p.seq_base = seq_base;
  p.seq_pad = seq_pad;
  if (drop && drop->thresh) { p.dkey = drop->key; p.dthresh = drop->thresh; p.dscale = drop->scale; }
  dim3 grid(B, cdiv(H, 4));
#define GO(HD) UR_ATTN_LAUNCH(attn_last_fwd_kernel, HD, grid, dim3(256), 0, st, q_last, qkv, seq, p, ctx_last, lse_last, qpv)
  switch (p.hd) {
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    case 32: GO(32); break;
    default: GO(64); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int attn_last_bwd(const float* q_last, const float* qkv, const int* seq, const float* ctx_last, const float* dctx_last,
                  const float* lse_last, int B, int L, int d, int H, float* dq_last, float* dqkv, hipStream_t st,
                  const int* seq_base, const int* seq_pad, const DropSpec* drop) {
  ProfScope ps(PC_ATTN_BWD, st, 8.0 * B * (double)L * d);
  AttnDims p;
  int rc = make_dims(B, L, d, H, 1, &p);
  if (rc) return rc;
  p.seq_base = seq_base;
  p.seq_pad = seq_pad;
  if (drop && drop->thresh) { p.dkey = drop->key; p.dthresh = drop->thresh; p.dscale = drop->scale; }
  dim3 grid(B, cdiv(H, 4));
#define GO(HD) UR_ATTN_LAUNCH(attn_last_bwd_kernel, HD, grid, dim3(256), 0, st, q_last, qkv, seq, ctx_last, dctx_last, lse_last, p, dq_last, dqkv)
  switch (p.hd) {
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    case 32: GO(32); break;
    default: GO(64); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

}  // namespace ur
