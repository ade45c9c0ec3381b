// fp32 MFMA GEMMs for the SASRec / GRU dense contractions (v_mfma_f32_32x32x2_f32: exact fp32,
// bit-equal to an fmaf chain -- MI355X_MICROARCH.md "Matrix cores").
//
//   gemm_nt : C[M,N] = epi( pro(A)[M,K] @ W[N,K]^T )        nn.Linear forward and, with a pre-transposed
//             weight copy, the activation-gradient GEMM dX = dY @ W.
//   gemm_tn : Out[R,Cc] = P[T,R]^T @ pro(Q)[T,Cc]            weight gradient dW = dY^T @ X (+ bias grad), several products per
//             launch (gemm_tn_group), split over the token dimension T with a deterministic second-stage reduction.
//
// Tiling: 256 threads = 4 waves in a 2x2 arrangement, each wave owns TM x TN MFMA tiles of 32x32.
// Operands are staged global -> registers -> LDS (double buffered, one barrier per K-step); LDS rows
// are padded to BK+4 floats so that the ds_read_b128 fragment reads are bank-conflict free.
// Epilogue: the accumulator tile goes through LDS (the staging buffers are free by then) so that every
// global access of the epilogue -- C, the residual / pre-activation operand, xhat -- is a coalesced
// 16-byte-per-lane row access; writing the MFMA fragment layout directly costs 64 dword stores per lane
// and made the store tail as long as the K loop.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace ur {

typedef float floatx16 __attribute__((ext_vector_type(16)));


__device__ __forceinline__ float4 act4(float4 v, int act) {
  v.x = act_fwd(v.x, act); v.y = act_fwd(v.y, act); v.z = act_fwd(v.z, act); v.w = act_fwd(v.w, act);
  return v;
}

// Epilogue shared by the GEMM kernels: the BM x BN accumulator tile sits in LDS (row stride BN+4); every global
// access -- C, the residual / pre-activation operand, xhat -- is a coalesced 16-byte-per-lane row access.
template <int BM, int BN, int EPI>
__device__ __forceinline__ void epilogue_from_lds(const float* Cs, int m0, int n0, int tid, const GemmArgs& a) {
  constexpr int CS = BN + 4;
  if constexpr (EPI == EPI_ADD_LNBWD) {
    // t = acc (+ aux): gradient wrt the LayerNorm output.  One 32-lane group per row (N <= 128: one float4 per lane), same
    // arithmetic as ln_bwd_kernel; the per-thread (d gamma, d beta) shares are summed over the tile's rows in a fixed order.
    const int g = tid >> 5, t = tid & 31;
    const int n4 = a.N >> 2;
    const float inv_n = 1.0f / (float)a.N;
    const bool cin = t < n4;
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), dg = gm, db = gm;
    if (cin) gm = *(const float4*)(a.gamma + t * 4);
    for (int ml = g; ml < BM; ml += 8) {
      const int m = m0 + ml;
      if (m >= a.M) break;
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f), h = y;
      if (cin) {
        y = *(const float4*)(Cs + ml * CS + t * 4);
        if (a.aux) {
          const float4 rs = *(const float4*)(a.aux + (long long)m * a.ldaux + t * 4);
          y.x += rs.x; y.y += rs.y; y.z += rs.z; y.w += rs.w;
        }
        h = *(const float4*)(a.xhat + (long long)m * a.N + t * 4);
      }
      dg.x += y.x * h.x; dg.y += y.y * h.y; dg.z += y.z * h.z; dg.w += y.w * h.w;
      db.x += y.x; db.y += y.y; db.z += y.z; db.w += y.w;
      float4 gy;
      gy.x = y.x * gm.x; gy.y = y.y * gm.y; gy.z = y.z * gm.z; gy.w = y.w * gm.w;
      const float m1 = group_sum<32>((gy.x + gy.y) + (gy.z + gy.w)) * inv_n;
      const float m2 = group_sum<32>((gy.x * h.x + gy.y * h.y) + (gy.z * h.z + gy.w * h.w)) * inv_n;
      const float r = a.rstd[m];
      if (cin) {
        float4 o;
        o.x = r * (gy.x - m1 - h.x * m2); o.y = r * (gy.y - m1 - h.y * m2);
        o.z = r * (gy.z - m1 - h.z * m2); o.w = r * (gy.w - m1 - h.w * m2);
        const long long orow = a.out_rows ? a.out_rows[m] : m;
        *(float4*)(a.C + orow * a.ldc + t * 4) = o;
      }
    }
    __syncthreads();   // every read of the staged accumulators is done: the buffer becomes the reduction scratch [8][2][CS]
    float* red = const_cast<float*>(Cs);
    if (cin) {
      *(float4*)(red + (g * 2 + 0) * CS + t * 4) = dg;
      *(float4*)(red + (g * 2 + 1) * CS + t * 4) = db;
    }
    __syncthreads();
    float* part = a.ln_part + (long long)(m0 / BM) * 2 * a.N;
    for (int i = tid; i < 2 * a.N; i += 256) {
      const int which = i / a.N, col = i % a.N;
      float acc = 0.f;
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) acc += red[(gg * 2 + which) * CS + col];
      part[i] = acc;
    }
  } else if constexpr (EPI != EPI_BIAS_RES_LN) {
    constexpr int RT = BN / 4;         // threads per row (one float4 each)
    constexpr int RPP = 256 / RT;      // rows per pass
    const int t = tid % RT, g = tid / RT;
    const int n = n0 + t * 4;
    if (n < a.N || EPI == EPI_COUNT_GT) {  // N % 4 == 0 (COUNT_GT requires N % BN == 0 so every lane is in range)
      float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
      if (EPI == EPI_BIAS) bias = *(const float4*)(a.bias + n);
#pragma unroll 4
      for (int ml = g; ml < BM; ml += RPP) {
        const int m = m0 + ml;
        if (m >= a.M) break;
        float4 v = *(const float4*)(Cs + ml * CS + t * 4);
        if (EPI == EPI_BIAS) { v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w; }
        if (EPI == EPI_MUL_DACT) {
          const float4 h = *(const float4*)(a.aux + (long long)m * a.ldaux + n);
          v.x *= act_bwd(h.x, a.act); v.y *= act_bwd(h.y, a.act); v.z *= act_bwd(h.z, a.act); v.w *= act_bwd(h.w, a.act);
        }
        if (EPI == EPI_ADD) {
          const float4 h = *(const float4*)(a.aux + (long long)m * a.ldaux + n);
          v.x += h.x; v.y += h.y; v.z += h.z; v.w += h.w;
          if (a.aux2) {
            const float4 h2 = *(const float4*)(a.aux2 + (long long)m * a.ldaux2 + n);
            v.x += h2.x; v.y += h2.y; v.z += h2.z; v.w += h2.w;
          }
          if (a.out_rows) {   // scatter-accumulate: row m is ADDED onto row out_rows[m] of C (distinct rows; replaces a scatter-add launch)
            float* dst = a.C + (long long)a.out_rows[m] * a.ldc + n;
            const float4 old = *(const float4*)dst;
            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
            *(float4*)dst = v;
            continue;
          }
        }
        if (EPI == EPI_COUNT_GT) {
          // full-item ranking: nothing is stored; count the columns of this tile whose score (acc + bias[n]) beats the
          // row's threshold aux[m], reduce over the RT lanes that share the row, one integer atomic per (row, tile)
          float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.bias) bs = *(const float4*)(a.bias + n);
          const float th = a.aux[m];
          const long long sk = (a.skip ? a.skip[m] - a.skip_base : -1) - n;   // column of this float4 to leave out (0..3) or none
          float cnt = ((v.x + bs.x > th && sk != 0) ? 1.f : 0.f) + ((v.y + bs.y > th && sk != 1) ? 1.f : 0.f) +
                      ((v.z + bs.z > th && sk != 2) ? 1.f : 0.f) + ((v.w + bs.w > th && sk != 3) ? 1.f : 0.f);
          cnt = group_sum<RT>(cnt);
          if (t == 0 && cnt > 0.f) atomicAdd((int*)a.C + m, (int)cnt);
          continue;
        }
        *(float4*)(a.C + (long long)m * a.ldc + n) = v;
      }
    }
  } else {
    // t = acc + bias + residual; one 32-lane group per row computes LayerNorm and writes y, xhat, rstd.
    constexpr int NV = BN >= 128 ? BN / 128 : 1;
    const int g = tid >> 5, t = tid & 31;
    const int n4 = a.N >> 2;
    const float inv_n = 1.0f / (float)a.N;
    for (int ml = g; ml < BM; ml += 8) {
      const int m = m0 + ml;
      if (m >= a.M) break;
      float4 v[NV];
      float s = 0.f;
      const unsigned rk = a.drop.thresh ? drop_rowkey(a.drop, m) : 0u;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = t + 32 * k;
        if (c < n4) {
          float4 x = *(const float4*)(Cs + ml * CS + c * 4);
          const float4 bs = *(const float4*)(a.bias + c * 4);
          const float4 rs = *(const float4*)(a.aux + (long long)m * a.ldaux + c * 4);
          if (a.drop.thresh) {   // t = dropout(acc + bias) + res
            x.x += bs.x; x.y += bs.y; x.z += bs.z; x.w += bs.w;
            x = drop4(x, rk, (unsigned)(c * 4), a.drop);
            x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w;
          } else {
            x.x += bs.x + rs.x; x.y += bs.y + rs.y; x.z += bs.z + rs.z; x.w += bs.w + rs.w;
          }
          v[k] = x;
          s += (x.x + x.y) + (x.z + x.w);
        }
      }
      const float mean = group_sum<32>(s) * inv_n;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = t + 32 * k;
        if (c < n4) {
          v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
          q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
      }
      const float rstd = 1.0f / sqrtf(group_sum<32>(q) * inv_n + a.eps);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = t + 32 * k;
        if (c < n4) {
          const float4 gm = *(const float4*)(a.gamma + c * 4), bt = *(const float4*)(a.beta + c * 4);
          float4 h, o;
          h.x = v[k].x * rstd; h.y = v[k].y * rstd; h.z = v[k].z * rstd; h.w = v[k].w * rstd;
          o.x = h.x * gm.x + bt.x; o.y = h.y * gm.y + bt.y; o.z = h.z * gm.z + bt.z; o.w = h.w * gm.w + bt.w;
          *(float4*)(a.xhat + (long long)m * a.N + c * 4) = h;
          *(float4*)(a.C + (long long)m * a.ldc + c * 4) = o;
        }
      }
      if (t == 0) a.rstd[m] = rstd;
    }
  }
}

// split-bf16 arithmetic (see gemm_tn_split_kernel below): the three bf16 pieces of the eight fp32 values of one lane's fragment of a
// 16-wide K slice (k offsets fk .. fk+3 and 8 + fk .. 8 + fk+3: the fragment layout of v_mfma_f32_32x32x16_bf16 with the k order
// permuted the same way on both operands); 9 VALU instructions per pair of values
typedef __bf16 nt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float nt_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int nt_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_split8(const float4 a0, const float4 a1, nt_bf16x8 (&pc)[3]) {
  const nt_f32x2 x[4] = {{a0.x, a0.y}, {a0.z, a0.w}, {a1.x, a1.y}, {a1.z, a1.w}};
  nt_u32x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const nt_f32x2 r = x[j] - __builtin_bit_cast(nt_f32x2, __builtin_bit_cast(nt_u32x2, x[j]) & 0xFFFF0000u);
    const nt_f32x2 t = r - __builtin_bit_cast(nt_f32x2, __builtin_bit_cast(nt_u32x2, r) & 0xFFFF0000u);
    h[j] = __builtin_amdgcn_perm(__float_as_uint(x[j][1]), __float_as_uint(x[j][0]), 0x07060302u);
    m[j] = __builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u);
    l[j] = __builtin_amdgcn_perm(__float_as_uint(t[1]), __float_as_uint(t[0]), 0x07060302u);
  }
  pc[0] = __builtin_bit_cast(nt_bf16x8, h); pc[1] = __builtin_bit_cast(nt_bf16x8, m); pc[2] = __builtin_bit_cast(nt_bf16x8, l);
}

// SPL (round 6d): the K loop in split-bf16 arithmetic -- both fragments split in registers where they are consumed, six piece products per
// (row block, column block, 16-wide slice) on v_mfma_f32_32x32x16_bf16 (6 x 8 passes where the fp32-input MFMA takes 8 x 16).  Staging,
// LDS image and epilogues are the exact kernel's.
template <int BM, int BN, int PRO, int EPI, int BK = 32, bool SPL = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs a) {
  constexpr int LS = BK + 4;            // padded LDS row stride (floats): conflict-free ds_read_b128 for BK = 16 and 32
  constexpr int C4N = BK / 4;           // float4 columns per tile row
  constexpr int RPT = 256 / C4N;        // tile rows covered per pass of the 256 threads
  constexpr int WM = (BM >= 64 && BM % 64 == 0) ? 2 : 1, WN = 4 / WM;    // wave grid (BM = 32 / 96: all four waves side by side along N)
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int AV = BM / RPT, WV = BN / RPT;  // float4 loads per thread per K-step
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM*LS]
  float* Ws = smem + 2 * BM * LS;   // [2][BN*LS]

  if (a.m_dev) a.M = min(a.M, *a.m_dev);   // compacted token rows: the row count lives on the device
  if (a.ksplit > 1) {   // this grid row's piece of K: [kb, kb + ks), a multiple of BK long; its partial product goes to its own copy of C
    const int ks = ((a.K + a.ksplit - 1) / a.ksplit + BK - 1) / BK * BK, kb = (int)blockIdx.y * ks;
    a.A += kb; a.W += kb;
    a.K = max(0, min(a.K - kb, ks));
    a.C += (long long)blockIdx.y * a.split_stride;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  // XCD-aware tile mapping: workgroup id b runs on XCD b % 8 (each XCD has its own L2).  All N-tiles of one M-tile are
  // given ids congruent mod 8, so the A tile they share is fetched into ONE L2 instead of one per N-tile
  // (rocprof FETCH_SIZE on the QKV GEMM: 41 MB -> 14 MB).
  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
  const int mt = (qid / ntn) * 8 + xcd, nt_ = qid % ntn;
  if (mt >= ntm) {
    if constexpr (EPI == EPI_ADD_LNBWD) {   // a surplus M-tile (compacted rows: fewer rows than the grid was sized for): zero partial sums
      if (mt < (a.M_host + BM - 1) / BM)
        for (int i = threadIdx.x; i < 2 * a.N; i += 256) a.ln_part[(long long)mt * 2 * a.N + i] = 0.f;
    }
    return;
  }
  const int m0 = mt * BM, n0 = nt_ * BN;
  const int c4 = tid % C4N, lrow = tid / C4N;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // out-of-range rows are clamped (their results are never stored); only the K tail needs zero fill
  const float* Ap[AV];
  const float* Wp[WV];
#pragma unroll
  for (int i = 0; i < AV; ++i) Ap[i] = a.A + (long long)min(m0 + lrow + RPT * i, a.M - 1) * a.lda + c4 * 4;
#pragma unroll
  for (int i = 0; i < WV; ++i) Wp[i] = a.W + (long long)min(n0 + lrow + RPT * i, a.N - 1) * a.ldw + c4 * 4;

  float4 ra[AV], rw[WV];
  auto load_global = [&](int kt) {
    const int k = kt * BK;
    if (k + c4 * 4 < a.K) {
#pragma unroll
      for (int i = 0; i < AV; ++i) ra[i] = *(const float4*)(Ap[i] + k);
#pragma unroll
      for (int i = 0; i < WV; ++i) rw[i] = *(const float4*)(Wp[i] + k);
    } else {
#pragma unroll
      for (int i = 0; i < AV; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < WV; ++i) rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      float4 v = ra[i];
      if (PRO == PRO_ACT) v = act4(v, a.act);
      *(float4*)(As + buf * BM * LS + (lrow + RPT * i) * LS + c4 * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) *(float4*)(Ws + buf * BN * LS + (lrow + RPT * i) * LS + c4 * 4) = rw[i];
  };

  const int nk = (a.K + BK - 1) / BK;
  const int frow = lane & 31, fk = 4 * (lane >> 5);
  load_global(0);
  store_lds(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_global(kt + 1);   // the next tile is in flight during this one's MFMAs
      const float* Ab = As + buf * BM * LS + (wr * (BM / WM) + frow) * LS + fk;
      const float* Wb = Ws + buf * BN * LS + (wc * (BN / WN) + frow) * LS + fk;
      if constexpr (SPL) {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
          nt_bf16x8 ap[TM][3], bp[TN][3];
#pragma unroll
          for (int i = 0; i < TM; ++i) nt_split8(*(const float4*)(Ab + i * 32 * LS + kk), *(const float4*)(Ab + i * 32 * LS + kk + 8), ap[i]);
#pragma unroll
          for (int j = 0; j < TN; ++j) nt_split8(*(const float4*)(Wb + j * 32 * LS + kk), *(const float4*)(Wb + j * 32 * LS + kk + 8), bp[j]);
          // small terms first, the leading product last (gemm_tn_split_kernel's order)
          constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
          for (int tm = 0; tm < 6; ++tm)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][PA[tm]], bp[j][PB[tm]], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        float4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const float4*)(Ab + i * 32 * LS + kk);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *(const float4*)(Wb + j * 32 * LS + kk);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
          }
      }
      }
      if (kt + 1 < nk) store_lds(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue.  acc[r]: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 inside the 32x32 tile.
  constexpr int CS = BN + 4;
  float* Cs = smem;  // [BM][CS] -- staging buffers are dead after the loop's final barrier
  {
    const int lcol = lane & 31, lrow4 = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nl = wc * (BN / WN) + j * 32 + lcol;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = wr * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + lrow4;
          Cs[ml * CS + nl] = acc[i][j][r];
        }
      }
  }
  __syncthreads();
  epilogue_from_lds<BM, BN, EPI>(Cs, m0, n0, tid, a);
}

template <int BM, int BN, int PRO, int EPI, int BK = 32, bool SPL = false>
static int launch_nt(const GemmArgs& a, hipStream_t st) {
  constexpr int LS = BK + 4;
  const long long nblk = 8LL * cdiv(cdiv(a.M, BM), 8) * cdiv(a.N, BN);
  if (nblk * 256 >= (1LL << 32)) return fail(UR_ERR_UNSUPPORTED, "gemm_nt: %lld workgroups exceed HIP's 2^32-thread grid limit", nblk);
  dim3 grid((unsigned)nblk, (unsigned)(a.ksplit > 1 ? a.ksplit : 1));
  size_t lds = (size_t)2 * (BM + BN) * LS * sizeof(float);
  const size_t cs = (size_t)BM * (BN + 4) * sizeof(float);
  if (cs > lds) lds = cs;
  static const hipError_t attr = hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, PRO, EPI, BK, SPL>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  UR_LAUNCH_EV((gemm_nt_kernel<BM, BN, PRO, EPI, BK, SPL>), grid, dim3(256), lds, st, a);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// Few rows (the last-row layer, the GRU's per-step GEMM): 64-row tiles leave most CUs idle and each workgroup MFMA-bound
// on its own K loop (M = 512, N = 128, K = 512: 8 workgroups x 13.7 us of MFMA).  32-row tiles double the workgroups.
static bool small_m(const GemmArgs& a) { return a.M <= 1024; }

// the arithmetic of the plain-epilogue products (dispatch_tile): split bf16 (six terms) when mfma_arith names a split form -- inside an
// encoder call its cfg's field, for the raw hook ur_gemm_nt the process-wide setting (test hook nt_split=0: exact whatever it says).
// The LayerNorm-epilogue launches and the ranking count (integer results compared bit for bit) stay on the fp32-input MFMA.
static bool nt_split_on() {
  static const bool hook = ur_test_hook("nt_split", 1) != 0;
  const int base = mfma_arith() & 0xFF;
  return hook && (base == 6 || base == 9);
}
#define UR_NT_GO(BM_, BN_, BK_) (spl ? launch_nt<BM_, BN_, PRO, EPI, BK_, true>(a, st) : launch_nt<BM_, BN_, PRO, EPI, BK_, false>(a, st))
template <int PRO, int EPI>
static int dispatch_tile(const GemmArgs& a, hipStream_t st) {
  const bool spl = EPI != EPI_COUNT_GT && nt_split_on();
  // enough 128x128 tiles to fill 256 CUs twice? otherwise use 64-row tiles for more workgroups
  const long long big = (long long)cdiv(a.M, 128) * cdiv(a.N, 128);
  if (a.N <= 64) return UR_NT_GO(64, 64, 32);
  if (small_m(a)) return UR_NT_GO(32, 128, 32);
  // compacted rows and a one-tile-wide output: the 64-row grid (M/64 workgroups, ~1.5 per CU) hides the rows that were
  // skipped behind wave quantisation; 64 x 64 tiles (as many workgroups as 32 x 128, 16 KB instead of 20 KB of operands per K-step)
  // let the saving through
  if (a.m_dev && a.N <= 128 && a.N > 64) return UR_NT_GO(64, 64, 32);
  if (a.m_dev && a.N <= 128) return UR_NT_GO(32, 128, 32);
  // short K, wide N (QKV, FFN-1, d-act): the 16-deep K-step variant keeps 4 workgroups per CU resident and measured
  // 6-10 % faster at M = 25600; elsewhere the 32-deep step wins
  if (a.K <= 128 && a.N >= 256) return UR_NT_GO(64, 128, 16);
  if (big >= 512) return UR_NT_GO(128, 128, 32);
  return UR_NT_GO(64, 128, 32);
}
#undef UR_NT_GO

// The launches whose epilogue needs whole rows (LayerNorm forward / backward: BN = 128 = one row) use 32-row tiles (64- / 96- / 128-row
// tiles measured slower at the C5 shapes: round 2, profiles/r02_c_ln_tile_rows.txt).
int gemm_nt_lnbwd_tiles(int M) { return cdiv(M, 32); }

int gemm_nt(const GemmArgs& a, int pro, int epi, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return UR_OK;
  ProfScope ps(PC_GEMM_NT, st, 2.0 * a.M * a.N * a.K, true);   // (one launch, through UR_LAUNCH_EV: the kernel's own timestamps)
  if ((a.K & 3) || (a.N & 3) || (a.lda & 3) || (a.ldw & 3) || (epi != EPI_COUNT_GT && ((a.ldc & 3) || (a.aux && (a.ldaux & 3)) || (a.aux2 && (a.ldaux2 & 3)))))
    return fail(UR_ERR_ARG, "gemm_nt: N, K and all leading dimensions must be multiples of 4 (N=%d K=%d)", a.N, a.K);
  if (a.ksplit > 1 && (epi != EPI_NONE || pro != PRO_NONE || a.m_dev || a.ksplit > 64))
    return fail(UR_ERR_UNSUPPORTED, "gemm_nt: a split K dimension needs the plain epilogue (epi=%d pro=%d ksplit=%d)", epi, pro, a.ksplit);
  if (epi == EPI_BIAS_RES_LN) {
    if (a.N > 256 || a.ldc != a.N) return fail(UR_ERR_UNSUPPORTED, "gemm_nt: fused LayerNorm needs N<=256 (N=%d)", a.N);
    if (a.N <= 128 && (small_m(a) || a.m_dev))
      return pro == PRO_ACT ? launch_nt<32, 128, PRO_ACT, EPI_BIAS_RES_LN>(a, st)
                            : launch_nt<32, 128, PRO_NONE, EPI_BIAS_RES_LN>(a, st);
    if (a.N <= 128) return pro == PRO_ACT ? launch_nt<64, 128, PRO_ACT, EPI_BIAS_RES_LN>(a, st)
                                          : launch_nt<64, 128, PRO_NONE, EPI_BIAS_RES_LN>(a, st);
    return pro == PRO_ACT ? launch_nt<64, 256, PRO_ACT, EPI_BIAS_RES_LN>(a, st)
                          : launch_nt<64, 256, PRO_NONE, EPI_BIAS_RES_LN>(a, st);
  }
  if (epi == EPI_ADD_LNBWD) {
    if (a.N > 128 || pro != PRO_NONE || !a.xhat || !a.rstd || !a.gamma || !a.ln_part)
      return fail(UR_ERR_UNSUPPORTED, "gemm_nt: fused LayerNorm backward needs N <= 128 and xhat / rstd / gamma / ln_part (N=%d)", a.N);
    const_cast<GemmArgs&>(a).M_host = a.M;
    return launch_nt<32, 128, PRO_NONE, EPI_ADD_LNBWD>(a, st);
  }
  if (pro == PRO_ACT) {
    if (epi == EPI_BIAS) return dispatch_tile<PRO_ACT, EPI_BIAS>(a, st);
    return fail(UR_ERR_UNSUPPORTED, "gemm_nt: PRO_ACT with epilogue %d", epi);
  }
  switch (epi) {
    case EPI_NONE: return dispatch_tile<PRO_NONE, EPI_NONE>(a, st);
    case EPI_BIAS: return dispatch_tile<PRO_NONE, EPI_BIAS>(a, st);
    case EPI_MUL_DACT: return dispatch_tile<PRO_NONE, EPI_MUL_DACT>(a, st);
    case EPI_ADD: return dispatch_tile<PRO_NONE, EPI_ADD>(a, st);
    case EPI_COUNT_GT:
      if (a.N % 128) return fail(UR_ERR_ARG, "gemm_nt: EPI_COUNT_GT needs N %% 128 == 0 (N=%d)", a.N);
      return launch_nt<128, 128, PRO_NONE, EPI_COUNT_GT>(a, st);
  }
  return fail(UR_ERR_UNSUPPORTED, "gemm_nt: epilogue %d", epi);
}

// ======================================================================================== gemm_tn
// Weight-gradient products Out_i[R_i, C_i] = P_i[T_i, R_i]^T pro(Q_i)[T_i, C_i] (+ bias gradient colsum(P_i)), SEVERAL per launch.
// Rounds 1-2 gave every product its own launch and filled the chip by splitting the token dimension ~86 ways: 9-11 launches per
// backward pass, each workgroup ran 8 LDS stages between a cold prologue and a 64 KB partial-tile store, and the deferred reduction
// read 85-100 MB of partial tiles back (1.77 x the algorithmic HBM bytes; 0.17 of the fp32-MFMA roof in situ).  Here the chip is
// filled ACROSS products: every (product, 64 x 64 output tile, token split) is one workgroup of a single grid, so the products queued
// at one fork of the backward pass need 8-24 token splits each; the partial tiles shrink to a few MB per launch, and a product whose T
// is small (the B last rows) takes S = 1 and writes its result (and bias gradient) directly.  The second stage of the split is the
// deferred ReduceBatch (fixed order: bit-reproducible).
//   workgroup = 4 waves (2 x 2), wave = one 32 x 32 accumulator; stage = 32 tokens x (64 + 64) columns, double buffered (32 KB);
//   odd token rows are stored with column bit 5 flipped (the two half-waves of a fragment read hit different banks);
//   out-of-range token rows load a zero row (pointer select: a select on the loaded VALUE makes the compiler branch around the load and
//   wait for it on the spot); fragments are read four steps ahead of their MFMA; the bias gradient is accumulated from the staging
//   registers (no LDS reads) by the workgroups of tile column 0.
constexpr int GT = 64;    // output tile edge
constexpr int GBT = 32;   // tokens per stage
struct TnItem {
  const float *P, *Q; int ldp, ldq;
  int T; const int* t_dev;       // t_dev (nullable): device-side token count, T = min(T, *t_dev)
  int R, Cc, pro_act, act;
  float *part, *bias_part;       // S > 1: [S][R * Cc] and [S][R] (nullable) partial results
  float *out, *bias_out; int ldo;   // S == 1: the result itself
  int S, ntr, ntc, first_block;
};
struct TnGroup { static constexpr int MAX = 12; TnItem item[MAX]; int n; int pro_prio; };

__global__ __launch_bounds__(256) void gemm_tn_group_kernel(TnGroup g, const float* __restrict__ zero_row) {
  int j = 0;
  while (j + 1 < g.n && (int)blockIdx.x >= g.item[j + 1].first_block) ++j;
  const TnItem& it = g.item[j];
  const int local = blockIdx.x - it.first_block, ntiles = it.ntr * it.ntc, S = it.S;
  int sp, tile;
  if ((S & 7) == 0) {   // all tiles of a split on one XCD (first_block % 8 == 0): its token rows enter one L2 only
    const int xcd = local & 7, qid = local >> 3;
    sp = (qid / ntiles) * 8 + xcd; tile = qid % ntiles;
  } else { sp = local % S; tile = local / S; }
  if (sp >= S || tile >= ntiles) return;
  int T = it.T;
  if (it.t_dev) T = min(T, *it.t_dev);
  const int tps = (((T + S - 1) / S + GBT - 1) / GBT) * GBT;
  const int t_begin = sp * tps, t_end = min(T, t_begin + tps);
  const int R = it.R, Cc = it.Cc, ldp = it.ldp, ldq = it.ldq, act = it.act;
  const bool pro = it.pro_act != 0;
  const int r0 = (tile / it.ntc) * GT, c0 = (tile % it.ntc) * GT;
  const bool want_bias = (it.bias_part != nullptr || (S == 1 && it.bias_out != nullptr)) && (tile % it.ntc) == 0;

  __shared__ __attribute__((aligned(16))) float smem[4 * GBT * GT];   // Ps[2][32*64] | Qs[2][32*64]; epilogue: Cs[64][68]
  float* Ps = smem;
  float* Qs = smem + 2 * GBT * GT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int c4 = tid & 15, trow = tid >> 4;   // 16 token rows per pass, 2 passes per stage
  const bool rin = r0 + c4 * 4 < R, cin = c0 + c4 * 4 < Cc;
  const float* Pp = it.P + (rin ? r0 + c4 * 4 : 0);
  const float* Qp = it.Q + (cin ? c0 + c4 * 4 : 0);
  typedef float tfx4 __attribute__((ext_vector_type(4)));
  tfx4 rp[2], rq[2];
  tfx4 bs = {0.f, 0.f, 0.f, 0.f};
  auto load_global = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = t0 + trow + 16 * i;
      const bool tin = t < t_end;
      const int tt = max(0, min(t, T - 1));
      const float* pp = (tin && rin) ? Pp + (long long)t * ldp : zero_row;
      rp[i] = *(const tfx4*)pp;
      rq[i] = *(const tfx4*)(cin ? Qp + (long long)tt * ldq : zero_row);
    }
  };
  auto store_lds = [&](int buf) {
    if (pro) {   // (one wave-uniform switch per stage, not one per element)
#define UR_ACT8(A_)                                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int e = 0; e < 4; ++e) rq[i][e] = act_fwd(rq[i][e], A_)
      switch (act) {
        case UR_ACT_GELU: UR_ACT8(UR_ACT_GELU); break;
        case UR_ACT_RELU: UR_ACT8(UR_ACT_RELU); break;
        case UR_ACT_SWISH: UR_ACT8(UR_ACT_SWISH); break;
        case UR_ACT_TANH: UR_ACT8(UR_ACT_TANH); break;
        case UR_ACT_SIGMOID: UR_ACT8(UR_ACT_SIGMOID); break;
        default: break;
      }
#undef UR_ACT8
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = trow + 16 * i;
      const int col = (c4 * 4) ^ ((row & 1) << 5);
      *(tfx4*)(Ps + buf * GBT * GT + row * GT + col) = rp[i];
      bs += rp[i];
      *(tfx4*)(Qs + buf * GBT * GT + row * GT + col) = rq[i];
    }
  };
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nt = (t_end - t_begin + GBT - 1) / GBT;
  if (nt > 0) {
    load_global(t_begin);
    store_lds(0);
  }
  __syncthreads();
  const int fcol = lane & 31, ft = lane >> 5;
  const int sw = ft << 5;
  const int pa = ft * GT + ((wr * 32 + fcol) ^ sw), qa = ft * GT + ((wc * 32 + fcol) ^ sw);
  for (int s_ = 0; s_ < nt; ++s_) {
    const int buf = s_ & 1;
    if (s_ + 1 < nt) load_global(t_begin + (s_ + 1) * GBT);
    const float* Pb = Ps + buf * GBT * GT + pa;
    const float* Qb = Qs + buf * GBT * GT + qa;
    constexpr int FD = 4;   // fragments read FD steps ahead of their MFMA
    float fa[FD], fb[FD];
#pragma unroll
    for (int u = 0; u < FD; ++u) { fa[u] = Pb[2 * u * GT]; fb[u] = Qb[2 * u * GT]; }
#pragma unroll
    for (int kk = 0; kk < GBT; kk += 2) {
      const int u = (kk >> 1) % FD;
      const float a = fa[u], b = fb[u];
      if (kk + 2 * FD < GBT) { fa[u] = Pb[(kk + 2 * FD) * GT]; fb[u] = Qb[(kk + 2 * FD) * GT]; }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s_ + 1 < nt) store_lds(buf ^ 1);
    __syncthreads();
  }
  // ---- epilogue through LDS: coalesced float4 row stores of the (partial) tile
  constexpr int CS = GT + 4;
  float* Cs = smem;
  {
    const int lrow4 = 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) Cs[(wr * 32 + (r & 3) + 8 * (r >> 2) + lrow4) * CS + wc * 32 + fcol] = acc[r];
  }
  float* Bs = smem + GT * CS;   // [16][64] bias partial sums of the 16 row groups
  if (want_bias) *(tfx4*)(Bs + trow * GT + c4 * 4) = bs;
  __syncthreads();
  const bool direct = S == 1;
  float* out = direct ? it.out : it.part + (long long)sp * R * Cc;
  const int ldo = direct ? it.ldo : Cc;
  {
    const int c = c0 + c4 * 4;
    if (c < Cc)
      for (int rl = trow; rl < GT; rl += 16) {
        const int rr = r0 + rl;
        if (rr >= R) break;
        *(float4*)(out + (long long)rr * ldo + c) = *(const float4*)(Cs + rl * CS + c4 * 4);
      }
  }
  if (want_bias && tid < GT && r0 + tid < R) {
    float b = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) b += Bs[k * GT + tid];
    if (direct) it.bias_out[r0 + tid] = b;
    else it.bias_part[(long long)sp * R + r0 + tid] = b;
  }
}

// ======================================================================================== gemm_tn on the bf16 matrix pipes
// fp32-EQUIVALENT arithmetic on v_mfma_f32_32x32x16_bf16 (round 6).  Every fp32 operand value is written as the EXACT sum of three
// bf16 pieces, x = x1 + x2 + x3 (truncation split: x1 = the upper 16 bits of x, x2 = the upper 16 bits of x - x1, x3 = x - x1 - x2, which
// has at most 8 significant bits left: 3 x 8 = 24 mantissa bits), and the product sum_t p q is accumulated in fp32 from the SIX piece
// products of order <= 2^-16 (p1q1, p1q2, p2q1, p2q2, p1q3, p3q1); the three dropped ones are <= 2^-24 relative, i.e. one fp32
// rounding of the exact product.  Piece products are exact in fp32 (8 x 8 bits), a K = 16 instruction rounds the accumulator once where
// the fp32-input MFMA rounds 8 times.  NTERM = 9 keeps all nine.  The bf16 pipe runs 16 x the fp32-input MFMA rate, so six
// instructions cost 6/16 of the exact kernel's matrix time (ceiling 2500 / 6 = 417 TFLOP/s fp32-equivalent against 157.3).
//   workgroup = 8 waves (2 x 4) on a 128 x 128 output tile, wave = 64 x 32 = two 32 x 32 accumulators; two 60 KB workgroups per CU =
//   four waves per SIMD, so one workgroup's split (VALU) runs under the other's MFMAs;
//   stage = 32 tokens x (128 + 128) features: wave w stages the 32-feature slab w of (P | Q): lane = (token group of four g = lane & 7,
//   feature quad q = lane >> 3) loads 4 token rows x float4 (PF stages ahead), splits, and writes -- per feature and piece -- its 4
//   tokens as one ds_write_b64: the LDS image is [operand][piece][feature][32 tokens] bf16, i.e. TOKEN-major inside a feature row, which
//   is what the MFMA wants (lane l of an A / B fragment holds 8 consecutive k of row / column l & 31, k group l >> 5): every fragment is
//   one ds_read_b128.  Rows are padded to 80 B: conflict-free for the b128 reads (16-lane groups: 20 l mod 64 distinct) and the b64
//   writes (16-lane groups: 2 g + 80 q' dwords mod 32 distinct).
//   Inf / NaN operands give NaN (inf - inf in the split), where the exact kernel may give inf: such a step is skipped either way.
constexpr int ST = 128;    // output tile edge of the wide form (d = 128-class products); the narrow form (TS = 64: d = 64 models) below
constexpr int SBT = 32;    // tokens per stage
constexpr int SRS = 80;    // bytes per LDS row (32 bf16 + 16 B pad)
constexpr int S_STAGE_BYTES = 6 * ST * SRS;   // 61 440 B (TS = 128); 30 720 B at TS = 64
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// The body of one workgroup, specialised on the term count and on the activation applied to Q (ACT < 0: none): the hot loop then has no
// data-dependent branch at all -- full stages only; a split's ragged last stage is peeled off and staged with clamped, zeroed rows.
// TS = tile edge: 128 = 8 waves (2 x 4), a wave = 64 x 32 (two accumulators); 64 = 4 waves (2 x 2), a wave = 32 x 32 (one): the same
// staging map (a wave = one 32-feature slab of P or Q, 16 elements per lane and stage), half the MFMAs per staged element.
template <int NTERM, int ACT, int TS>
__device__ __forceinline__ void tn_split_wg(const TnItem& it, const int sp, const int tile, const float* __restrict__ zero_row,
                                            unsigned char* smem, long long* trace) {
  constexpr int SW = TS / 32;            // staging waves per operand = waves across the tile's columns
  constexpr int MI = TS / 64;            // 32-row accumulator blocks per wave (the wave grid is 2 x SW)
  int T = it.T;
  if (it.t_dev) T = min(T, *it.t_dev);
  const int S = it.S;
  const int tps = (((T + S - 1) / S + SBT - 1) / SBT) * SBT;
  const int t_begin = sp * tps, t_end = min(T, t_begin + tps);
  const int R = it.R, Cc = it.Cc;
  const int r0 = (tile / it.ntc) * TS, c0 = (tile % it.ntc) * TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave / SW, wc = wave % SW;
  // ---- staging role: waves 0-3 stage P, 4-7 stage Q; a wave = a 32-feature slab; lane = (token group of four sg = lane & 7, feature
  // quad lane >> 3): four token rows x float4 per stage
  const int so = wave / SW, sg = lane & 7, sf0 = (wave % SW) * 32 + (lane >> 3) * 4;
  const int sdim = so ? Cc : R, sorg = so ? c0 : r0;
  const bool fin = sorg + sf0 < sdim;
  // a feature quad beyond the operand's width reads the zero row at stride 0: no select per load
  const float* const sbase = fin ? (so ? it.Q : it.P) + sorg + sf0 : zero_row;
  const long long sld = fin ? (so ? it.ldq : it.ldp) : 0;
  const bool want_bias = so == 0 && (it.bias_part != nullptr || (S == 1 && it.bias_out != nullptr)) && (tile % it.ntc) == 0;
  typedef float tfx4 __attribute__((ext_vector_type(4)));
  tfx4 xa[4], xb[4];   // the stages in flight: loaded two stages before they are split
  tfx4 bs = {0.f, 0.f, 0.f, 0.f};
  const int n_tok = max(t_end - t_begin, 0);
  const int nfull = n_tok / SBT, ntail = n_tok - nfull * SBT, nt = nfull + (ntail ? 1 : 0);
  const float* sptr = sbase + (long long)(t_begin + 4 * sg) * sld;   // row of this lane's first token; moves by 32 rows a stage
  const long long sstep = (long long)SBT * sld;
  auto load_full = [&](tfx4 (&x)[4], int stage) {
    const float* p0 = sptr + (long long)stage * sstep;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *(const tfx4*)(p0 + (long long)i * sld);
  };
  auto load_tail = [&](tfx4 (&x)[4]) {   // the ragged last stage: rows clamped to the split's last token, zeroed behind it
    const int t0 = t_begin + nfull * SBT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + 4 * sg + i;
      x[i] = *(const tfx4*)(sbase + (long long)min(t, t_end - 1) * sld);
      if (t >= t_end) x[i] = tfx4{0.f, 0.f, 0.f, 0.f};
    }
  };
  unsigned char* const sdst = smem + (so * 3 * TS + sf0) * SRS + sg * 8;
  auto split_store = [&](tfx4 (&x)[4]) {
    if (ACT >= 0 && so) {   // (wave-uniform) the activation of the FFN's hidden rows, recomputed as in gemm_tn_group_kernel
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[i][e] = act_fwd(x[i][e], ACT);
    }
    if (want_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bs += x[i];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {   // feature sf0 + e: four consecutive tokens -> 8 bytes per piece
      float r[4], r2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = x[i][e] - __uint_as_float(__float_as_uint(x[i][e]) & 0xFFFF0000u);
#pragma unroll
      for (int i = 0; i < 4; ++i) r2[i] = r[i] - __uint_as_float(__float_as_uint(r[i]) & 0xFFFF0000u);
      u32x2 hi, mid, lo;
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // (odd token's upper half | even token's upper half)
        hi[h] = __builtin_amdgcn_perm(__float_as_uint(x[2 * h + 1][e]), __float_as_uint(x[2 * h][e]), 0x07060302u);
        mid[h] = __builtin_amdgcn_perm(__float_as_uint(r[2 * h + 1]), __float_as_uint(r[2 * h]), 0x07060302u);
        lo[h] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * h + 1]), __float_as_uint(r2[2 * h]), 0x07060302u);
      }
      *(u32x2*)(sdst + e * SRS) = hi;
      *(u32x2*)(sdst + (TS + e) * SRS) = mid;
      *(u32x2*)(sdst + (2 * TS + e) * SRS) = lo;
    }
  };
  floatx16 acc[MI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const unsigned char* const fa0 = smem + (wr * 32 * MI + (lane & 31)) * SRS + (lane >> 5) * 16;
  const unsigned char* const fb0 = smem + (3 * TS + wc * 32 + (lane & 31)) * SRS + (lane >> 5) * 16;
  auto compute = [&]() {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 fa[MI][3], fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fb[p] = *(const bf16x8*)(fb0 + (p * TS) * SRS + kb * 32);
#pragma unroll
        for (int m = 0; m < MI; ++m) fa[m][p] = *(const bf16x8*)(fa0 + (p * TS + m * 32) * SRS + kb * 32);
      }
      // small terms first, the leading product last
      constexpr int PA[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
      constexpr int PB[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int tm = 9 - NTERM; tm < 9; ++tm)
#pragma unroll
        for (int m = 0; m < MI; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][PA[tm]], fb[PB[tm]], acc[m], 0, 0, 0);
    }
  };
  int tr = 0;
  auto stamp = [&]() {
    if (trace && tid == 0 && tr < 62) trace[blockIdx.x * 64 + 2 + tr++] = (long long)__builtin_amdgcn_s_memtime();
  };
  // ---- prologue: stage 0 -> LDS; stages 1 and 2 on their way
  if (trace && tid == 0) trace[blockIdx.x * 64 + 62] = wall_clock64();   // (100 MHz, the same counter on every CU)
  stamp();
  if (nfull > 0) {
    load_full(xa, 0);
    if (nfull > 1) load_full(xb, 1);
    split_store(xa);
    if (nfull > 2) load_full(xa, 2);
  } else if (ntail) {
    load_tail(xa);
    split_store(xa);
  }
  __syncthreads();
  stamp();
  // ---- full stages, two per trip: stage s computes from LDS while s + 1 waits in registers (xb, then xa) and s + 2 / s + 3 are requested
  int s_ = 0;
  for (; s_ + 2 < nfull; s_ += 2) {
    compute();
    stamp();
    __syncthreads();
    split_store(xb);                                   // stage s + 1
    if (s_ + 3 < nfull) load_full(xb, s_ + 3);
    stamp();
    __syncthreads();
    compute();
    __syncthreads();
    split_store(xa);                                   // stage s + 2
    if (s_ + 4 < nfull) load_full(xa, s_ + 4);
    __syncthreads();
    stamp();
  }
  // ---- the last one or two full stages and the ragged tail
  if (s_ < nfull) {           // stage s_ is in LDS; s_ + 1 (if any) waits in xb
    compute();
    __syncthreads();
    if (s_ + 1 < nfull) {
      split_store(xb);
      __syncthreads();
      compute();
      __syncthreads();
    }
    if (ntail) {
      load_tail(xa);
      split_store(xa);
      __syncthreads();
      compute();
    }
  } else if (nfull == 0 && ntail) {
    compute();
  }
  stamp();
  // ---- epilogue: a store instruction writes two 128-byte row pieces (lanes 0-31 | 32-63)
  const bool direct = S == 1;
  float* out = direct ? it.out : it.part + (long long)sp * R * Cc;
  const int ldo = direct ? it.ldo : Cc;
  const int c = c0 + wc * 32 + (lane & 31);
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = r0 + wr * 32 * MI + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (rr < R && c < Cc) out[(long long)rr * ldo + c] = acc[m][r];
    }
  if (want_bias) {   // (wave-uniform: so == 0) column sums of P: the eight token groups of a feature quad are eight adjacent lanes
    bs.x = group_sum<8>(bs.x); bs.y = group_sum<8>(bs.y); bs.z = group_sum<8>(bs.z); bs.w = group_sum<8>(bs.w);
    if (sg == 0 && fin) {
      float* bo = direct ? it.bias_out : it.bias_part + (long long)sp * R;
      *(tfx4*)(bo + r0 + sf0) = bs;
    }
  }
  stamp();
  if (trace && tid == 0) { trace[blockIdx.x * 64] = tr; trace[blockIdx.x * 64 + 1] = nt; trace[blockIdx.x * 64 + 63] = wall_clock64(); }
}

template <int NTERM, int TS>
__global__ __launch_bounds__(TS * 4, 4) void gemm_tn_split_kernel(TnGroup g, const float* __restrict__ zero_row, long long* trace) {
  int j = 0;
  while (j + 1 < g.n && (int)blockIdx.x >= g.item[j + 1].first_block) ++j;
  const TnItem& it = g.item[j];
  const int local = blockIdx.x - it.first_block, ntiles = it.ntr * it.ntc, S = it.S;
  int sp, tile;
  if ((S & 7) == 0) {
    const int xcd = local & 7, qid = local >> 3;
    sp = (qid / ntiles) * 8 + xcd; tile = qid % ntiles;
  } else { sp = local % S; tile = local / S; }
  if (sp >= S || tile >= ntiles) return;
  __shared__ __attribute__((aligned(16))) unsigned char smem[6 * TS * SRS];
  if (!it.pro_act) { tn_split_wg<NTERM, -1, TS>(it, sp, tile, zero_row, smem, trace); return; }
  if (g.pro_prio) __builtin_amdgcn_s_setprio(1);
  switch (it.act) {   // (one specialised loop per activation: the workgroup runs exactly one of them)
    case UR_ACT_GELU: tn_split_wg<NTERM, UR_ACT_GELU, TS>(it, sp, tile, zero_row, smem, trace); break;
    case UR_ACT_RELU: tn_split_wg<NTERM, UR_ACT_RELU, TS>(it, sp, tile, zero_row, smem, trace); break;
    case UR_ACT_SWISH: tn_split_wg<NTERM, UR_ACT_SWISH, TS>(it, sp, tile, zero_row, smem, trace); break;
    case UR_ACT_TANH: tn_split_wg<NTERM, UR_ACT_TANH, TS>(it, sp, tile, zero_row, smem, trace); break;
    case UR_ACT_SIGMOID: tn_split_wg<NTERM, UR_ACT_SIGMOID, TS>(it, sp, tile, zero_row, smem, trace); break;
    default: tn_split_wg<NTERM, -1, TS>(it, sp, tile, zero_row, smem, trace); break;
  }
}

// a few zero floats in device memory (per device, allocated once): what out-of-range token rows load
static const float* tn_zero_buf() {
  static float* z[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!z[dev]) {
    float* p = nullptr;
    if (hipMalloc((void**)&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    z[dev] = p;
  }
  return z[dev];
}

constexpr int TN_GROUP_SMAX = 32;         // token splits of a product, exact-fp32 kernel (64 x 64 tiles)
constexpr int TN_SPLIT_SMAX = 48;         // ... split-bf16 kernel (128 x 128 tiles: a quarter of the tiles, so more splits fill the chip)
long long gemm_tn_group_ws_floats(int R, int Cc) { return (long long)TN_SPLIT_SMAX * ((long long)R * Cc + R) + 64; }

// req[i].ws: gemm_tn_group_ws_floats(R, Cc) floats each (untouched until the deferred reduction has run).  defer == nullptr: the
// reduction of the split products runs right behind the launch.
// ---- which arithmetic the weight-gradient products run in: 0 = exact fp32 MFMA (the default), 6 / 9 = bf16x6 / bf16x9 split on the
// bf16 pipes (fp32-equivalent; gemm_tn_split_kernel), 3 = a three-term split (NARROWER than fp32: error studies only).
// Initial value: test hook tn_split=<n>; ur_set_mfma_arith overrides it.
static std::atomic<int> g_mfma_arith{-1};
static thread_local int g_arith_scope = -1;   // >= 0: the arithmetic of the encoder call in progress (UrSasrecCfg / UrGruCfg .mfma_arith)
ArithScope::ArithScope(int terms) : prev(g_arith_scope) {
  const int base = terms & 0xFF;
  g_arith_scope = (base == 6 || base == 9 || base == 3) ? terms & 0x3FF : 0;
}
ArithScope::~ArithScope() { g_arith_scope = prev; }
int mfma_arith() {
  if (g_arith_scope >= 0) return g_arith_scope;
  int m = g_mfma_arith.load(std::memory_order_relaxed);
  if (m < 0) {
    m = ur_test_hook("tn_split", 0);
    if (m != 3 && m != 6 && m != 9) m = 0;
    g_mfma_arith.store(m, std::memory_order_relaxed);
  }
  return m;
}
int set_mfma_arith(int m) {
  if (m != 0 && m != 3 && m != 6 && m != 9) return fail(UR_ERR_ARG, "mfma_arith: %d (0 = exact fp32, 6 / 9 = split bf16 terms)", m);
  g_mfma_arith.store(m, std::memory_order_relaxed);
  return UR_OK;
}

static int gemm_tn_group_split(const TnReq* req, int n, hipStream_t st, ReduceBatch* defer, int nterm, const float* zeros, int ts) {
  // two 60 KB workgroups per CU; a workgroup should walk >= 8 stages between its cold prologue and its 64 KB partial-tile store
  // (the narrow form's workgroups are 4 waves and 30 KB: four of them fit where two wide ones do)
  const int target = ur_test_hook("tn_split_target", ts == ST ? 512 : 1024);
  // a stage of a product whose Q operand takes the activation costs its staging waves ~1/3 more (measured: profiles/r06_*_tn_split_trace):
  // such a product gets proportionally more, shorter splits, so that the workgroups of a launch end together
  const double pro_cost = ur_test_hook("tn_split_procost", 135) * 0.01;
  TnGroup g{};
  g.n = n;
  g.pro_prio = ur_test_hook("tn_split_proprio", 0);
  double work = 0.0, flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const TnReq& q = req[i];
    if ((q.R & 3) || (q.Cc & 3) || (q.ldp & 3) || (q.ldq & 3) || (q.ldo & 3)) return fail(UR_ERR_ARG, "gemm_tn: R/Cc/ld must be multiples of 4");
    if (q.T <= 0) return fail(UR_ERR_ARG, "gemm_tn: T=%d", q.T);
    work += (double)cdiv(q.R, ts) * cdiv(q.Cc, ts) * q.T * (q.pro_act ? pro_cost : 1.0);
    flops += 2.0 * q.T * q.R * q.Cc;
  }
  const double rows_per = std::max(256.0, work / target);
  int blocks = 0;
  ReduceBatch local;
  ReduceBatch* rb = defer ? defer : &local;
  for (int i = 0; i < n; ++i) {
    const TnReq& q = req[i];
    TnItem& it = g.item[i];
    int S = (int)(q.T * (q.pro_act ? pro_cost : 1.0) / rows_per + 0.5);
    if (S > q.T / (2 * SBT)) S = q.T / (2 * SBT);
    if (S >= 6) S = std::min(TN_SPLIT_SMAX, (S + 4) / 8 * 8);
    if (S < 1) S = 1;
    it.P = q.P; it.Q = q.Q; it.ldp = q.ldp; it.ldq = q.ldq; it.T = q.T; it.t_dev = q.t_dev; it.R = q.R; it.Cc = q.Cc;
    it.pro_act = q.pro_act; it.act = q.act; it.S = S; it.ntr = cdiv(q.R, ts); it.ntc = cdiv(q.Cc, ts);
    it.out = q.out; it.bias_out = q.bias_out; it.ldo = q.ldo;
    it.part = q.ws; it.bias_part = q.bias_out ? q.ws + (long long)S * q.R * q.Cc : nullptr;
    it.first_block = blocks;
    blocks += (S & 7) == 0 ? S * it.ntr * it.ntc : 8 * cdiv(S * it.ntr * it.ntc, 8);
  }
  {
    ProfScope ps(PC_GEMM_TN, st, flops, true);
    long long* trace = nullptr;
    static const int want_trace = ur_test_hook("tn_split_trace", 0);
    static long long* trace_buf = nullptr;
    if (want_trace) {
      if (!trace_buf && hipMalloc((void**)&trace_buf, 4096 * 64 * sizeof(long long)) != hipSuccess) trace_buf = nullptr;
      trace = blocks <= 4096 ? trace_buf : nullptr;
    }
#define UR_TNS(N_) do { if (ts == ST) { UR_LAUNCH_EV((gemm_tn_split_kernel<N_, 128>), dim3(blocks), dim3(512), 0, st, g, zeros, trace); } \
                       else { UR_LAUNCH_EV((gemm_tn_split_kernel<N_, 64>), dim3(blocks), dim3(256), 0, st, g, zeros, trace); } } while (0)
    if (nterm == 9) UR_TNS(9);
    else if (nterm == 3) UR_TNS(3);
    else UR_TNS(6);
#undef UR_TNS
    if (trace && want_trace == 2) {   // debugging aid: print the phase stamps of a few workgroups of THIS launch
      static int printed = 0;
      if (printed++ == 3 && hipStreamSynchronize(st) == hipSuccess) {
        std::vector<long long> h((size_t)blocks * 64);
        if (hipMemcpy(h.data(), trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
          long long w0 = -1, w1 = 0;
          for (int b = 0; b < blocks; ++b) if (h[b * 64] > 0) { if (w0 < 0 || h[b * 64 + 62] < w0) w0 = h[b * 64 + 62]; w1 = std::max(w1, h[b * 64 + 63]); }
          fprintf(stderr, "tn_split trace: %d workgroups, first start -> last end %.2f us (100 MHz wall clock)\n", blocks, (w1 - w0) * 0.01);
          for (int b = 0; b < blocks; b += std::max(1, blocks / 24)) {
            const int n = (int)h[b * 64];
            long long cyc = 0;
            for (int k = 1; k < n; ++k) cyc += h[b * 64 + 2 + k] - h[b * 64 + 2 + k - 1];
            fprintf(stderr, "  wg %4d nt %2lld  starts at %6.2f us, lives %6.2f us = %7lld shader ticks; prologue %lld, first trips", b, h[b * 64 + 1],
                    (h[b * 64 + 62] - w0) * 0.01, (h[b * 64 + 63] - h[b * 64 + 62]) * 0.01, cyc, n > 1 ? h[b * 64 + 3] - h[b * 64 + 2] : 0);
            for (int k = 2; k < std::min(n, 8); ++k) fprintf(stderr, " %lld", h[b * 64 + 2 + k] - h[b * 64 + 2 + k - 1]);
            fprintf(stderr, " ... last");
            for (int k = std::max(8, n - 5); k < n; ++k) fprintf(stderr, " %lld", h[b * 64 + 2 + k] - h[b * 64 + 2 + k - 1]);
            fprintf(stderr, "\n");
          }
        }
      }
    }
  }
  UR_LAUNCH_CHECK();
  for (int i = 0; i < n; ++i) {
    const TnItem& it = g.item[i];
    if (it.S == 1) continue;
    const long long ne = (long long)it.R * it.Cc;
    if (rb->full(2)) {
      int rc = reduce_batch(*rb, st);
      if (rc) return rc;
    }
    rb->add(it.part, ne, it.S, ne, it.Cc, it.out, it.ldo);
    if (it.bias_out) rb->add(it.bias_part, it.R, it.S, it.R, it.R, it.bias_out, it.R);
  }
  if (!defer) return reduce_batch(local, st);
  return UR_OK;
}

int gemm_tn_group(const TnReq* req, int n, hipStream_t st, ReduceBatch* defer) {
  if (n <= 0) return UR_OK;
  if (n > TnGroup::MAX) return fail(UR_ERR_ARG, "gemm_tn_group: %d products (max %d)", n, TnGroup::MAX);
  const float* zeros = tn_zero_buf();
  if (!zeros) return fail(UR_ERR_HIP, "gemm_tn: no device memory for the zero row");
  if (const int arith = mfma_arith()) {
    // the split kernel has a wide (128 x 128 tiles, 8 waves) and a narrow (64 x 64, 4 waves) form; products that would fill less than
    // 70 % of even the narrow tiles keep the exact kernel inside an encoder call unless the cfg says 0x100 | terms ("every shape": the
    // unit tests); the raw hooks (ur_gemm_tn / ur_gemm_tn_group under ur_set_mfma_arith) take a split form at every shape
    double used = 0.0, tiled = 0.0, tiled64 = 0.0;
    for (int i = 0; i < n; ++i) {
      used += (double)req[i].R * req[i].Cc * req[i].T;
      tiled += (double)cdiv(req[i].R, ST) * cdiv(req[i].Cc, ST) * ST * ST * req[i].T;
      tiled64 += (double)cdiv(req[i].R, 64) * cdiv(req[i].Cc, 64) * 64 * 64 * req[i].T;
    }
    const bool any = g_arith_scope < 0 || (arith & 0x100);
    static const int force_ts = ur_test_hook("tn_split_ts", 0);
    if (force_ts == 64 || force_ts == 128) return gemm_tn_group_split(req, n, st, defer, arith & 0xFF, zeros, force_ts);
    // (the wide form needs a partner workgroup on its CU -- one's split runs under the other's MFMAs: a group of few tiles cannot be cut
    // into ~2 workgroups per CU (C4's encoder at H = 128: two products, six tiles, measured 8 % slower than the exact kernel) and stays exact;
    // scope bit 0x200 -- set by ur_sasrec_bwd -- waives that: its small group (the last-row layer's K, V product + B-row products) runs
    // on the side stream UNDER the backward row chain, where the shorter matrix-pipe time matters and the fill does not: - 4.5 us per step)
    long long tiles128 = 0;
    for (int i = 0; i < n; ++i) tiles128 += (long long)cdiv(req[i].R, ST) * cdiv(req[i].Cc, ST) * (req[i].T >= 2048 ? 1 : 0);
    static const int min_wg = ur_test_hook("tn_split_minwg", 384);
    const bool enough = any || (arith & 0x200) || tiles128 * TN_SPLIT_SMAX >= min_wg;
    if (used >= 0.7 * tiled && enough) return gemm_tn_group_split(req, n, st, defer, arith & 0xFF, zeros, ST);     // d = 128-class products
    if (used >= 0.7 * tiled && !enough) goto exact;
    if (any || used >= 0.7 * tiled64) return gemm_tn_group_split(req, n, st, defer, arith & 0xFF, zeros, 64);   // d = 64-class (round 6b)
  }
exact:
  // workgroups per launch (~3 per CU: 32 KB of LDS each).  Measured at C5 (profiles/r03_a_dw_schedule.txt): 288 -> 0.681 ms/step, 576 ->
  // 0.660, 864 -> 0.658, 1152+ -> 0.665 -- short workgroups give the CUs back to the main stream's kernels sooner
  constexpr int target = 864;
  TnGroup g{};
  g.n = n;
  double work = 0.0, flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const TnReq& q = req[i];
    if ((q.R & 3) || (q.Cc & 3) || (q.ldp & 3) || (q.ldq & 3) || (q.ldo & 3)) return fail(UR_ERR_ARG, "gemm_tn: R/Cc/ld must be multiples of 4");
    if (q.T <= 0) return fail(UR_ERR_ARG, "gemm_tn: T=%d", q.T);
    work += (double)cdiv(q.R, GT) * cdiv(q.Cc, GT) * q.T;
    flops += 2.0 * q.T * q.R * q.Cc;
  }
  const double rows_per = std::max(128.0, work / target);   // token rows one workgroup walks
  int blocks = 0;
  ReduceBatch local;
  ReduceBatch* rb = defer ? defer : &local;
  for (int i = 0; i < n; ++i) {
    const TnReq& q = req[i];
    TnItem& it = g.item[i];
    int S = (int)(q.T / rows_per + 0.5);
    if (S > q.T / (2 * GBT)) S = q.T / (2 * GBT);
    if (S >= 6) S = std::min(TN_GROUP_SMAX, (S + 4) / 8 * 8);
    if (S < 1) S = 1;
    it.P = q.P; it.Q = q.Q; it.ldp = q.ldp; it.ldq = q.ldq; it.T = q.T; it.t_dev = q.t_dev; it.R = q.R; it.Cc = q.Cc;
    it.pro_act = q.pro_act; it.act = q.act; it.S = S; it.ntr = cdiv(q.R, GT); it.ntc = cdiv(q.Cc, GT);
    it.out = q.out; it.bias_out = q.bias_out; it.ldo = q.ldo;
    it.part = q.ws; it.bias_part = q.bias_out ? q.ws + (long long)S * q.R * q.Cc : nullptr;
    it.first_block = blocks;
    blocks += (S & 7) == 0 ? S * it.ntr * it.ntc : 8 * cdiv(S * it.ntr * it.ntc, 8);   // (every product starts on XCD 0)
  }
  {
    ProfScope ps(PC_GEMM_TN, st, flops, true);   // (the launch's own timestamps: on the side stream a recorded bracket would hold the waits of every fork)
    UR_LAUNCH_EV(gemm_tn_group_kernel, dim3(blocks), dim3(256), 0, st, g, zeros);
  }
  UR_LAUNCH_CHECK();
  for (int i = 0; i < n; ++i) {
    const TnItem& it = g.item[i];
    if (it.S == 1) continue;
    const long long ne = (long long)it.R * it.Cc;
    if (rb->full(2)) {
      int rc = reduce_batch(*rb, st);
      if (rc) return rc;
    }
    rb->add(it.part, ne, it.S, ne, it.Cc, it.out, it.ldo);
    if (it.bias_out) rb->add(it.bias_part, it.R, it.S, it.R, it.R, it.bias_out, it.R);
  }
  if (!defer) return reduce_batch(local, st);
  return UR_OK;
}

long long gemm_tn_ws_floats(int T, int R, int Cc) { (void)T; return gemm_tn_group_ws_floats(R, Cc); }

// ---- all queued split reductions in one launch.  Block = 16 float4 columns x 16 slices: slice k sums the partials
// s = k, k+16, ...; the 16 slice sums are combined through LDS in slice order (fixed order: bit-reproducible).
__global__ __launch_bounds__(256) void reduce_batch_kernel(ReduceBatch rb) {
  __shared__ float4 red[16][16];
  int j = 0;
  while (j + 1 < rb.n && (int)blockIdx.x >= rb.item[j + 1].first_block) ++j;
  const ReduceItem it = rb.item[j];
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long long i = ((long long)(blockIdx.x - it.first_block) * 16 + c) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < it.n) {
    const float* p = it.part + i;
#pragma unroll 4
    for (int s = sl; s < it.S; s += 16) {
      const float4 v = *(const float4*)(p + (long long)s * it.stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[sl][c] = acc;
  __syncthreads();
  if (sl == 0 && i < it.n) {
    float4 t = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = red[k][c];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *(float4*)(it.out + (i / it.cols) * it.ldo + (i % it.cols)) = t;
  }
}

int reduce_batch(ReduceBatch& rb0, hipStream_t st) {
  for (ReduceBatch* b = &rb0; b; b = b->next) {
    ReduceBatch& rb = *b;
    if (rb.n <= 0) continue;
    ProfScope ps(PC_GEMM_TN, st, 0.0);
    int blocks = 0;
    for (int i = 0; i < rb.n; ++i) {
      rb.item[i].first_block = blocks;
      blocks += cdiv(rb.item[i].n / 4, 16);
    }
    hipLaunchKernelGGL(reduce_batch_kernel, dim3(blocks), dim3(256), 0, st, rb);
    UR_LAUNCH_CHECK();
    rb.n = 0;
  }
  return UR_OK;
}

// one product = a group of one (same kernel, same split rule)
int gemm_tn(const float* P, int ldp, const float* Q, int ldq, int T, int R, int Cc, int pro_act_on_q, int act, float* out,
            int ldo, float* bias_out, float* ws, hipStream_t st, ReduceBatch* defer, const int* t_dev) {
  const TnReq q{P, ldp, Q, ldq, T, R, Cc, pro_act_on_q, act, out, ldo, bias_out, ws, t_dev};
  return gemm_tn_group(&q, 1, st, defer);
}

// ------------------------------------------------------------------------------------- transpose
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int rows, int cols,
                                                        float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (by + j < rows && bx + tx < cols) tile[j][tx] = src[(long long)(by + j) * cols + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < cols && by + tx < rows) dst[(long long)(bx + j) * rows + by + tx] = tile[tx][j];
}

int transpose(const float* src, int rows, int cols, float* dst, hipStream_t st) {
  ProfScope ps(PC_MISC, st, 8.0 * rows * cols);
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, st, src, rows, cols, dst);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// several transposes in ONE launch (all weight matrices of the encoder before its backward pass: 8 launches -> 1)
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch tb) {
  __shared__ float tile[32][33];
  if (tb.copy_src && blockIdx.x == 0 && threadIdx.x == 0) *tb.copy_dst = *tb.copy_src;
  if (tb.zero2_ptr && (int)blockIdx.x >= tb.zero2_first_block) {   // riders: zero-fill (the backward's gradient buffers)
    const long long i0 = ((long long)(blockIdx.x - tb.zero2_first_block) * 256 + threadIdx.x) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long e = i0 + q * 4;
      if (e >= tb.zero2_n) continue;
      if (tb.zero2_pad) {   // only the padded positions of every sequence: the valid rows are written whole by their producer
        const long long row = e / tb.zero2_d;
        if ((int)(row % tb.zero2_L) >= tb.zero2_pad[row / tb.zero2_L]) continue;
      }
      *(float4*)(tb.zero2_ptr + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  if (tb.zero_ptr && (int)blockIdx.x >= tb.zero_first_block) {
    const long long i0 = ((long long)(blockIdx.x - tb.zero_first_block) * 256 + threadIdx.x) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i0 + q * 4 < tb.zero_n) *(float4*)(tb.zero_ptr + i0 + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  if (tb.n <= 0) return;   // (riders only)
  int j = 0;
  while (j + 1 < tb.n && (int)blockIdx.x >= tb.item[j + 1].first_block) ++j;
  const TransposeItem it = tb.item[j];
  if (it.mode) {   // split copy: a thread = (K-slice kb, k group g, column n) = one 16-byte cell of each of the three piece planes
    const int om = it.mode & 3, SL = (it.mode & 4) ? 32 : 16, NG = SL / 8;   // slice depth, k groups per slice
    const int K = om == 1 ? it.rows : it.cols, N = om == 1 ? it.cols : it.rows;
    const long long cell = (long long)(blockIdx.x - it.first_block) * 256 + threadIdx.x;
    if (cell >= (long long)(K / SL) * NG * N) return;
    const int n = (int)(cell % N), g = (int)(cell / N) % NG, kb = (int)(cell / ((long long)NG * N));
    // the eight k of the cell: lay 16 = {4 g .. 4 g + 3, 8 + 4 g ..} of the slice, lay 32 = 8 g .. 8 g + 7
    const int k0 = SL * kb + (SL == 16 ? 4 * g : 8 * g), k1 = SL == 16 ? k0 + 8 : k0 + 4;
    float x[8];
    if (om == 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = it.src[(long long)((q < 4 ? k0 : k1 - 4) + q) * N + n];
    } else {
      const float4 lo4 = *(const float4*)(it.src + (long long)n * K + k0), hi4 = *(const float4*)(it.src + (long long)n * K + k1);
      x[0] = lo4.x; x[1] = lo4.y; x[2] = lo4.z; x[3] = lo4.w; x[4] = hi4.x; x[5] = hi4.y; x[6] = hi4.z; x[7] = hi4.w;
    }
    typedef unsigned int tu32x4 __attribute__((ext_vector_type(4)));
    tu32x4 pc[3];
#pragma unroll
    for (int h = 0; h < 4; ++h) {   // (odd element's upper half | even element's upper half), as tn_split_wg::split_store
      const float xe = x[2 * h], xo = x[2 * h + 1];
      const float re = xe - __uint_as_float(__float_as_uint(xe) & 0xFFFF0000u), ro = xo - __uint_as_float(__float_as_uint(xo) & 0xFFFF0000u);
      const float se = re - __uint_as_float(__float_as_uint(re) & 0xFFFF0000u), so = ro - __uint_as_float(__float_as_uint(ro) & 0xFFFF0000u);
      pc[0][h] = __builtin_amdgcn_perm(__float_as_uint(xo), __float_as_uint(xe), 0x07060302u);
      pc[1][h] = __builtin_amdgcn_perm(__float_as_uint(ro), __float_as_uint(re), 0x07060302u);
      pc[2][h] = __builtin_amdgcn_perm(__float_as_uint(so), __float_as_uint(se), 0x07060302u);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) ((tu32x4*)it.dst)[((long long)(kb * 3 + q) * NG + g) * N + n] = pc[q];
    return;
  }
  const int t = blockIdx.x - it.first_block, tcols = (it.cols + 31) / 32;
  const int bx = (t % tcols) * 32, by = (t / tcols) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    if (by + r < it.rows && bx + tx < it.cols) tile[r][tx] = it.src[(long long)(by + r) * it.cols + bx + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (bx + r < it.cols && by + tx < it.rows) it.dst[(long long)(bx + r) * it.rows + by + tx] = tile[tx][r];
}

int transpose_batch(TransposeBatch& tb, hipStream_t st) {
  if (tb.n <= 0 && !tb.zero_ptr && !tb.zero2_ptr && !tb.copy_src) return UR_OK;
  ProfScope ps(PC_MISC, st, 0.0);
  int blocks = 0;
  for (int i = 0; i < tb.n; ++i) {
    tb.item[i].first_block = blocks;
    if (tb.item[i].mode) {
      const int om = tb.item[i].mode & 3, SL = (tb.item[i].mode & 4) ? 32 : 16;
      const int K = om == 1 ? tb.item[i].rows : tb.item[i].cols, N = om == 1 ? tb.item[i].cols : tb.item[i].rows;
      if (K % SL) return fail(UR_ERR_ARG, "transpose_batch: split copy of a matrix with K=%d", K);
      blocks += cdiv((long long)(K / 8) * N, 256);   // one thread per 8 k of a column
      continue;
    }
    blocks += cdiv(tb.item[i].cols, 32) * cdiv(tb.item[i].rows, 32);
  }
  if (tb.zero_ptr) {
    tb.zero_first_block = blocks;
    blocks += cdiv(tb.zero_n, 4096);
  }
  if (tb.zero2_ptr) {
    tb.zero2_first_block = blocks;
    blocks += cdiv(tb.zero2_n, 4096);
  }
  if (blocks == 0) blocks = 1;   // (only the one-int copy rides)
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(blocks), dim3(256), 0, st, tb);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

}  // namespace ur

// ---- raw entry points (unit tests and micro-benchmarks of the GEMM kernels; see include/unirec_amd.h)
extern "C" int ur_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, int pro,
                          int epi, int act, const float* bias, const float* aux, int ldaux, const float* gamma, const float* beta,
                          float eps, float* xhat, float* rstd, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, UR_ERR_ARG, "ur_gemm_nt: bad argument");
  ur::GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.act = act; g.bias = bias;
  g.aux = aux; g.ldaux = ldaux; g.gamma = gamma; g.beta = beta; g.eps = eps; g.xhat = xhat; g.rstd = rstd;
  return ur::gemm_nt(g, pro, epi, ur::as_stream(stream));
}
extern "C" int64_t ur_gemm_tn_workspace_floats(int T, int R, int Cc) { return ur::gemm_tn_ws_floats(T, R, Cc); }
extern "C" int ur_gemm_tn(const float* P, int ldp, const float* Q, int ldq, int T, int R, int Cc, int pro_act_on_q, int act,
                          float* out, int ldo, float* bias_out, float* ws, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(P && Q && out && ws, UR_ERR_ARG, "ur_gemm_tn: null pointer");
  return ur::gemm_tn(P, ldp, Q, ldq, T, R, Cc, pro_act_on_q, act, out, ldo, bias_out, ws, ur::as_stream(stream));
}
extern "C" int ur_gemm_tn_group(int n, const float* const* P, const int* ldp, const float* const* Q, const int* ldq, const int* T,
                                const int* R, const int* Cc, const int* pro_act_on_q, int act, float* const* out, const int* ldo,
                                float* const* bias_out, float* const* ws, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(n > 0 && n <= 12 && P && ldp && Q && ldq && T && R && Cc && out && ldo && ws, UR_ERR_ARG, "ur_gemm_tn_group: bad argument (1 <= n <= 12)");
  ur::TnReq rq[12];
  for (int i = 0; i < n; ++i) {
    UR_REQUIRE(P[i] && Q[i] && out[i] && ws[i], UR_ERR_ARG, "ur_gemm_tn_group: null pointer in product %d", i);
    rq[i] = ur::TnReq{P[i], ldp[i], Q[i], ldq[i], T[i], R[i], Cc[i], pro_act_on_q ? pro_act_on_q[i] : 0, act, out[i], ldo[i],
                      bias_out ? bias_out[i] : nullptr, ws[i], nullptr};
  }
  return ur::gemm_tn_group(rq, n, ur::as_stream(stream));
}
extern "C" int ur_set_mfma_arith(int terms) {
  UR_TRACE_SCOPE();
  return ur::set_mfma_arith(terms);
}
extern "C" int ur_get_mfma_arith(void) { return ur::mfma_arith(); }
