// Masked multi-head self-attention, forward and backward (unirec/model/modules.py:284-311 with the
// additive mask of unirec/model/sequential/sasrec.py:40-57).
//
// SASRec heads are tiny (n_heads=16 => head dim 4..8), so QK^T and P.V run on the VALU out of LDS and
// MFMA is reserved for the dense projections.  One workgroup owns one sequence and a
// group of heads: K, V (and Q, dO in the backward) of that head group live in LDS; one lane owns one
// (query row, head) pair in the row passes and one (key row, head) pair in the column pass, so every
// softmax reduction is lane-local and all lanes of a wave read the same K/V row (LDS broadcast).
// Nothing of size [B,h,L,L] is ever written: the backward recomputes P from Q, K and the saved
// log-sum-exp.
//
// Mask semantics are the reference's, literally: allowed(i,j) = item_seq[j] > 0 and (j <= i if causal).
// Rows with at least one allowed key skip the masked keys (their softmax weight underflows to exactly 0
// in fp32: exp(-10000 - max)).  Rows with NO allowed key (left padding, empty history) take the literal
// path: every key gets s/sqrt(hd) + (-10000.0f) and the softmax runs over all L keys, as torch does.
#include "common.h"
#include "kernels.h"

namespace ur {

struct AttnDims {
  int B, L, d, H, hd, HG, causal;
  float scale;    // 1/sqrt(hd)
  float sqrt_hd;  // sqrt(hd) for the literal (division) path
};

template <int HDP>
__device__ __forceinline__ float dotq(const float (&q)[HDP], const float* __restrict__ k) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HDP; ++c) s = fmaf(q[c], k[c], s);
  return s;
}

// Stage columns [col0, col0+HG*hd) of rows [b*L, b*L+L) of a [*, ld] matrix into lds[L][HG*HDP] (zero padded).
template <int HDP>
__device__ __forceinline__ void stage_heads(const float* __restrict__ src, int ld, int col0, int L, int HG, int hd,
                                            float* lds) {
  const int CW = HG * HDP;
  for (int idx = threadIdx.x; idx < L * CW; idx += blockDim.x) {
    const int j = idx / CW, r = idx % CW, h = r / HDP, c = r % HDP;
    lds[idx] = (c < hd) ? src[(long long)j * ld + col0 + h * hd + c] : 0.f;
  }
}

template <int HDP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ seq, AttnDims p,
                                                       float* __restrict__ ctx, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = p.L, d = p.d, hd = p.hd, HG = p.HG, CW = HG * HDP;
  float* Ks = smem;
  float* Vs = smem + L * CW;
  int* valid = (int*)(smem + 2 * L * CW);
  const int b = blockIdx.x, h0 = blockIdx.y * HG;
  const float* base = qkv + (long long)b * L * 3 * d;
  stage_heads<HDP>(base, 3 * d, d + h0 * hd, L, HG, hd, Ks);
  stage_heads<HDP>(base, 3 * d, 2 * d + h0 * hd, L, HG, hd, Vs);
  for (int j = threadIdx.x; j < L; j += blockDim.x) valid[j] = seq[(long long)b * L + j] > 0;
  __syncthreads();

  const int LQP = (L + 63) & ~63;
  for (int item0 = 0; item0 < HG * LQP; item0 += blockDim.x) {
    const int item = item0 + threadIdx.x;
    const int h = item / LQP, i = item % LQP;  // h is wave-uniform (LQP % 64 == 0)
    if (h >= HG || i >= L) continue;
    const int wave_i_max = min(L - 1, (__builtin_amdgcn_readfirstlane(item) % LQP) + 63);
    float q[HDP];
#pragma unroll
    for (int c = 0; c < HDP; ++c) q[c] = (c < hd) ? base[(long long)i * 3 * d + (h0 + h) * hd + c] : 0.f;
    const float* Kh = Ks + h * HDP;
    const float* Vh = Vs + h * HDP;
    const int jend = p.causal ? wave_i_max + 1 : L;
    float m = -INFINITY;
    int cnt = 0;
    for (int j = 0; j < jend; ++j) {
      if (!valid[j]) continue;
      if (p.causal && j > i) continue;
      m = fmaxf(m, dotq<HDP>(q, Kh + j * CW) * p.scale);
      ++cnt;
    }
    float l = 0.f, o[HDP];
#pragma unroll
    for (int c = 0; c < HDP; ++c) o[c] = 0.f;
    if (cnt > 0) {
      for (int j = 0; j < jend; ++j) {
        if (!valid[j]) continue;
        if (p.causal && j > i) continue;
        const float pj = __expf(dotq<HDP>(q, Kh + j * CW) * p.scale - m);
        l += pj;
#pragma unroll
        for (int c = 0; c < HDP; ++c) o[c] = fmaf(pj, Vh[j * CW + c], o[c]);
      }
    } else {  // literal path: every key masked
      for (int j = 0; j < L; ++j) m = fmaxf(m, dotq<HDP>(q, Kh + j * CW) / p.sqrt_hd + -10000.0f);
      for (int j = 0; j < L; ++j) {
        const float pj = __expf((dotq<HDP>(q, Kh + j * CW) / p.sqrt_hd + -10000.0f) - m);
        l += pj;
#pragma unroll
        for (int c = 0; c < HDP; ++c) o[c] = fmaf(pj, Vh[j * CW + c], o[c]);
      }
    }
    const float inv_l = 1.0f / l;
    float* out = ctx + ((long long)b * L + i) * d + (h0 + h) * hd;
#pragma unroll
    for (int c = 0; c < HDP; ++c)
      if (c < hd) out[c] = o[c] * inv_l;
    lse[((long long)b * p.H + h0 + h) * L + i] = m + __logf(l);
  }
}

// Backward. dqkv[:, 0:d] = dQ, [d:2d] = dK, [2d:3d] = dV.
template <int HDP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const int* __restrict__ seq,
                                                       const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                       const float* __restrict__ lse, AttnDims p, float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = p.L, d = p.d, hd = p.hd, HG = p.HG, CW = HG * HDP;
  float* Qs = smem;
  float* Ks = Qs + L * CW;
  float* Vs = Ks + L * CW;
  float* Gs = Vs + L * CW;        // dO
  float* Ls = Gs + L * CW;        // lse  [HG][L]
  float* Ds = Ls + HG * L;        // D_i  [HG][L] = sum_c dO[i,c] * O[i,c]
  int* valid = (int*)(Ds + HG * L);
  int* deg = valid + L;           // row i has no allowed key
  const int b = blockIdx.x, h0 = blockIdx.y * HG;
  const float* base = qkv + (long long)b * L * 3 * d;
  stage_heads<HDP>(base, 3 * d, h0 * hd, L, HG, hd, Qs);
  stage_heads<HDP>(base, 3 * d, d + h0 * hd, L, HG, hd, Ks);
  stage_heads<HDP>(base, 3 * d, 2 * d + h0 * hd, L, HG, hd, Vs);
  stage_heads<HDP>(dctx + (long long)b * L * d, d, h0 * hd, L, HG, hd, Gs);
  for (int idx = threadIdx.x; idx < HG * L; idx += blockDim.x) {
    const int h = idx / L, i = idx % L;
    Ls[idx] = lse[((long long)b * p.H + h0 + h) * L + i];
    const float* o = ctx + ((long long)b * L + i) * d + (h0 + h) * hd;
    const float* g = dctx + ((long long)b * L + i) * d + (h0 + h) * hd;
    float s = 0.f;
    for (int c = 0; c < hd; ++c) s = fmaf(o[c], g[c], s);
    Ds[idx] = s;
  }
  for (int j = threadIdx.x; j < L; j += blockDim.x) valid[j] = seq[(long long)b * L + j] > 0;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    int any = 0;
    const int jend = p.causal ? i + 1 : L;
    for (int j = 0; j < jend; ++j) any |= valid[j];
    deg[i] = !any;
  }
  __syncthreads();

  const int LQP = (L + 63) & ~63;
  // ---- row pass: dQ_i = scale * sum_j dS_ij K_j,  dS_ij = P_ij (dO_i . V_j - D_i)
  for (int item0 = 0; item0 < HG * LQP; item0 += blockDim.x) {
    const int item = item0 + threadIdx.x;
    const int h = item / LQP, i = item % LQP;
    if (h >= HG || i >= L) continue;
    const int wave_i_max = min(L - 1, (__builtin_amdgcn_readfirstlane(item) % LQP) + 63);
    float q[HDP], g[HDP], dq[HDP];
#pragma unroll
    for (int c = 0; c < HDP; ++c) {
      q[c] = Qs[i * CW + h * HDP + c];
      g[c] = Gs[i * CW + h * HDP + c];
      dq[c] = 0.f;
    }
    const float* Kh = Ks + h * HDP;
    const float* Vh = Vs + h * HDP;
    const float li = Ls[h * L + i], Di = Ds[h * L + i];
    if (!deg[i]) {
      const int jend = p.causal ? wave_i_max + 1 : L;
      for (int j = 0; j < jend; ++j) {
        if (!valid[j]) continue;
        if (p.causal && j > i) continue;
        const float pj = __expf(dotq<HDP>(q, Kh + j * CW) * p.scale - li);
        const float ds = pj * (dotq<HDP>(g, Vh + j * CW) - Di);
#pragma unroll
        for (int c = 0; c < HDP; ++c) dq[c] = fmaf(ds, Kh[j * CW + c], dq[c]);
      }
#pragma unroll
      for (int c = 0; c < HDP; ++c) dq[c] *= p.scale;
    } else {
      for (int j = 0; j < L; ++j) {
        const float pj = __expf((dotq<HDP>(q, Kh + j * CW) / p.sqrt_hd + -10000.0f) - li);
        const float ds = pj * (dotq<HDP>(g, Vh + j * CW) - Di);
#pragma unroll
        for (int c = 0; c < HDP; ++c) dq[c] = fmaf(ds, Kh[j * CW + c], dq[c]);
      }
#pragma unroll
      for (int c = 0; c < HDP; ++c) dq[c] /= p.sqrt_hd;
    }
    float* out = dqkv + ((long long)b * L + i) * 3 * d + (h0 + h) * hd;
#pragma unroll
    for (int c = 0; c < HDP; ++c)
      if (c < hd) out[c] = dq[c];
  }
  // ---- column pass: dK_j = sum_i dS_ij Q_i * scale,  dV_j = sum_i P_ij dO_i
  for (int item0 = 0; item0 < HG * LQP; item0 += blockDim.x) {
    const int item = item0 + threadIdx.x;
    const int h = item / LQP, j = item % LQP;
    if (h >= HG || j >= L) continue;
    const int wave_j_min = __builtin_amdgcn_readfirstlane(item) % LQP;
    float k[HDP], v[HDP], dk[HDP], dv[HDP];
#pragma unroll
    for (int c = 0; c < HDP; ++c) {
      k[c] = Ks[j * CW + h * HDP + c];
      v[c] = Vs[j * CW + h * HDP + c];
      dk[c] = 0.f;
      dv[c] = 0.f;
    }
    const float* Qh = Qs + h * HDP;
    const float* Gh = Gs + h * HDP;
    const bool vj = valid[j] != 0;
    const int ibeg = 0;  // degenerate rows attend to every key, so all rows are visited; masked ones are skipped below
    (void)wave_j_min;
    for (int i = ibeg; i < L; ++i) {
      const bool dg = deg[i] != 0;  // wave-uniform
      float pj;
      if (dg) {
        pj = __expf((dotq<HDP>(k, Qh + i * CW) / p.sqrt_hd + -10000.0f) - Ls[h * L + i]);
      } else {
        if (!vj || (p.causal && j > i)) continue;
        pj = __expf(dotq<HDP>(k, Qh + i * CW) * p.scale - Ls[h * L + i]);
      }
      const float ds = pj * (dotq<HDP>(v, Gh + i * CW) - Ds[h * L + i]) * (dg ? 1.0f / p.sqrt_hd : p.scale);
#pragma unroll
      for (int c = 0; c < HDP; ++c) {
        dk[c] = fmaf(ds, Qh[i * CW + c], dk[c]);
        dv[c] = fmaf(pj, Gh[i * CW + c], dv[c]);
      }
    }
    float* outk = dqkv + ((long long)b * L + j) * 3 * d + d + (h0 + h) * hd;
    float* outv = outk + d;
#pragma unroll
    for (int c = 0; c < HDP; ++c)
      if (c < hd) {
        outk[c] = dk[c];
        outv[c] = dv[c];
      }
  }
}

static int pick_hdp(int hd) {
  int p = 4;
  while (p < hd) p <<= 1;
  return p;
}

// largest divisor HG of H such that n_arrays * L * HG * HDP floats (+ small tails) fit the LDS budget
static int pick_hg(int H, int L, int hdp, int n_arrays, size_t budget_bytes) {
  int best = 0;
  for (int hg = 1; hg <= H; ++hg) {
    if (H % hg) continue;
    const size_t need = ((size_t)n_arrays * L * hg * hdp + 2 * (size_t)hg * L + 2 * (size_t)L) * sizeof(float);
    if (need <= budget_bytes) best = hg;
  }
  return best;
}

long long attn_lse_floats(int B, int H, int L) { return (long long)B * H * L; }

template <int HDP>
static int launch_fwd(const float* qkv, const int* seq, const AttnDims& p, float* ctx, float* lse, size_t lds, hipStream_t st) {
  static int max_set = 0;
  if ((int)lds > max_set) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<HDP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    max_set = (int)lds;
  }
  hipLaunchKernelGGL((attn_fwd_kernel<HDP>), dim3(p.B, p.H / p.HG), dim3(256), lds, st, qkv, seq, p, ctx, lse);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
template <int HDP>
static int launch_bwd(const float* qkv, const int* seq, const float* ctx, const float* dctx, const float* lse,
                      const AttnDims& p, float* dqkv, size_t lds, hipStream_t st) {
  static int max_set = 0;
  if ((int)lds > max_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<HDP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    max_set = (int)lds;
  }
  hipLaunchKernelGGL((attn_bwd_kernel<HDP>), dim3(p.B, p.H / p.HG), dim3(256), lds, st, qkv, seq, ctx, dctx, lse, p, dqkv);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

static int make_dims(int B, int L, int d, int H, int causal, int n_arrays, AttnDims* p, int* hdp, size_t* lds) {
  if (H <= 0 || d % H) return fail(UR_ERR_ARG, "attention: d=%d not divisible by n_heads=%d", d, H);
  const int hd = d / H;
  if (hd > 64) return fail(UR_ERR_UNSUPPORTED, "attention: head dim %d > 64 is not supported yet", hd);
  *hdp = pick_hdp(hd);
  int hg = pick_hg(H, L, *hdp, n_arrays, 60 * 1024);
  if (hg == 0) hg = pick_hg(H, L, *hdp, n_arrays, 160 * 1024 - 256);
  if (hg == 0) return fail(UR_ERR_UNSUPPORTED, "attention: L=%d with head dim %d does not fit LDS (key tiling not implemented)", L, hd);
  p->B = B; p->L = L; p->d = d; p->H = H; p->hd = hd; p->HG = hg; p->causal = causal;
  p->sqrt_hd = sqrtf((float)hd);
  p->scale = 1.0f / p->sqrt_hd;
  *lds = ((size_t)n_arrays * L * hg * (*hdp) + 2 * (size_t)hg * L + 2 * (size_t)L) * sizeof(float);
  return UR_OK;
}

int attn_fwd(const float* qkv, const int* seq, int B, int L, int d, int H, int causal, float* ctx, float* lse,
             int q_last_only, hipStream_t st) {
  if (q_last_only) return fail(UR_ERR_UNSUPPORTED, "attn_fwd: last-row mode not implemented");
  AttnDims p;
  int hdp;
  size_t lds;
  int rc = make_dims(B, L, d, H, causal, 2, &p, &hdp, &lds);
  if (rc) return rc;
  switch (hdp) {
    case 4: return launch_fwd<4>(qkv, seq, p, ctx, lse, lds, st);
    case 8: return launch_fwd<8>(qkv, seq, p, ctx, lse, lds, st);
    case 16: return launch_fwd<16>(qkv, seq, p, ctx, lse, lds, st);
    case 32: return launch_fwd<32>(qkv, seq, p, ctx, lse, lds, st);
    default: return launch_fwd<64>(qkv, seq, p, ctx, lse, lds, st);
  }
}

int attn_bwd(const float* qkv, const int* seq, const float* ctx, const float* dctx, const float* lse, int B, int L, int d,
             int H, int causal, float* dqkv, int q_last_only, hipStream_t st) {
  if (q_last_only) return fail(UR_ERR_UNSUPPORTED, "attn_bwd: last-row mode not implemented");
  AttnDims p;
  int hdp;
  size_t lds;
  int rc = make_dims(B, L, d, H, causal, 4, &p, &hdp, &lds);
  if (rc) return rc;
  switch (hdp) {
    case 4: return launch_bwd<4>(qkv, seq, ctx, dctx, lse, p, dqkv, lds, st);
    case 8: return launch_bwd<8>(qkv, seq, ctx, dctx, lse, p, dqkv, lds, st);
    case 16: return launch_bwd<16>(qkv, seq, ctx, dctx, lse, p, dqkv, lds, st);
    case 32: return launch_bwd<32>(qkv, seq, ctx, dctx, lse, p, dqkv, lds, st);
    default: return launch_bwd<64>(qkv, seq, ctx, dctx, lse, p, dqkv, lds, st);
  }
}

}  // namespace ur
