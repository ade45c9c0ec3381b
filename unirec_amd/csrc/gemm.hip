// fp32 MFMA GEMMs for the SASRec / GRU dense contractions (v_mfma_f32_32x32x2_f32: exact fp32,
// bit-equal to an fmaf chain -- MI355X_MICROARCH.md "Matrix cores").
//
//   gemm_nt : C[M,N] = epi( pro(A)[M,K] @ W[N,K]^T )        nn.Linear forward and, with a pre-transposed
//             weight copy, the activation-gradient GEMM dX = dY @ W.
//   gemm_tn : Out[R,Cc] = P[T,R]^T @ pro(Q)[T,Cc]            weight gradient dW = dY^T @ X (+ bias grad),
//             split over the token dimension T with a deterministic second-stage reduction.
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

// PF = global-load prefetch distance in K-steps (register slots).  1: the next tile is in flight during the current
// one (enough when several workgroups share a CU); 4: for launches with few workgroups (M <= 1024 rows), where nothing
// else hides the L2/HBM round trip of every K-step.
// PIPE = 1: software-pipelined K loop.  With one or two 32 x 32 accumulators per wave a K-step is 16-32 dependent MFMAs between two
// barriers, and the schedule above exposes, per step, the LDS round trip of the fragment reads, the global -> LDS hand-over and the
// HBM latency of operands that are not cache resident (one slice of loads in flight per workgroup).  Here the three are taken off
// the MFMA chain: a ring of THREE LDS stages (tile t is stored during step t-2, its fragments are read into a second register set
// during step t-1 -- underneath that step's MFMAs -- and multiplied during step t), and two register sets for the global loads, issued
// FOUR steps ahead.  Same K order, same results.
template <int BM, int BN, int PRO, int EPI, int BK = 32, int PF = 1, int PIPE = 0>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs a) {
  UR_PRIO_MAIN();
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

  float4 ra[PF][AV], rw[PF][WV];
  auto load_global = [&](int kt, int slot) {
    const int k = kt * BK;
    if (k + c4 * 4 < a.K) {
#pragma unroll
      for (int i = 0; i < AV; ++i) ra[slot][i] = *(const float4*)(Ap[i] + k);
#pragma unroll
      for (int i = 0; i < WV; ++i) rw[slot][i] = *(const float4*)(Wp[i] + k);
    } else {
#pragma unroll
      for (int i = 0; i < AV; ++i) ra[slot][i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < WV; ++i) rw[slot][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_lds = [&](int buf, int slot) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      float4 v = ra[slot][i];
      if (PRO == PRO_ACT) v = act4(v, a.act);
      *(float4*)(As + buf * BM * LS + (lrow + RPT * i) * LS + c4 * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) *(float4*)(Ws + buf * BN * LS + (lrow + RPT * i) * LS + c4 * 4) = rw[slot][i];
  };

  const int nk = (a.K + BK - 1) / BK;
  const int dbg = a.debug;
  const int frow = lane & 31, fk = 4 * (lane >> 5);
  if constexpr (PIPE) {
    constexpr int NKK = BK / 8;
    float* As3 = smem;                  // [3][BM*LS]
    float* Ws3 = smem + 3 * BM * LS;    // [3][BN*LS]
    float4 ga[2][AV], gw[2][WV];        // global-load register sets
    float4 fa[2][NKK][TM], fb[2][NKK][TN];
    auto load_g = [&](int kt, int set) {
      const int k = kt * BK;
      if (k + c4 * 4 < a.K) {
#pragma unroll
        for (int i = 0; i < AV; ++i) ga[set][i] = *(const float4*)(Ap[i] + k);
#pragma unroll
        for (int i = 0; i < WV; ++i) gw[set][i] = *(const float4*)(Wp[i] + k);
      } else {
#pragma unroll
        for (int i = 0; i < AV; ++i) ga[set][i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < WV; ++i) gw[set][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto store_r = [&](int ring, int set) {
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        float4 v = ga[set][i];
        if (PRO == PRO_ACT) v = act4(v, a.act);
        *(float4*)(As3 + ring * BM * LS + (lrow + RPT * i) * LS + c4 * 4) = v;
      }
#pragma unroll
      for (int i = 0; i < WV; ++i) *(float4*)(Ws3 + ring * BN * LS + (lrow + RPT * i) * LS + c4 * 4) = gw[set][i];
    };
    auto read_f = [&](int ring, int fs) {
      const float* Ab = As3 + ring * BM * LS + (wr * (BM / WM) + frow) * LS + fk;
      const float* Wb = Ws3 + ring * BN * LS + (wc * (BN / WN) + frow) * LS + fk;
#pragma unroll
      for (int q = 0; q < NKK; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[fs][q][i] = *(const float4*)(Ab + i * 32 * LS + q * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[fs][q][j] = *(const float4*)(Wb + j * 32 * LS + q * 8);
      }
    };
    load_g(0, 0);
    if (nk > 1) load_g(1, 1);
    store_r(0, 0);
    if (nk > 1) store_r(1, 1);
    if (nk > 2) load_g(2, 0);
    if (nk > 3) load_g(3, 1);
    __syncthreads();
    read_f(0, 0);
    int ring = 0;   // ring slot of tile kt
    for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kt = kt0 + u;
        if (kt >= nk) break;
        const int r1 = ring == 2 ? 0 : ring + 1, r2 = r1 == 2 ? 0 : r1 + 1;
        if (kt + 1 < nk) read_f(r1, u ^ 1);   // next step's fragments: in flight underneath this step's MFMAs
#pragma unroll
        for (int q = 0; q < NKK; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][q][i].x, fb[u][q][j].x, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][q][i].y, fb[u][q][j].y, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][q][i].z, fb[u][q][j].z, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][q][i].w, fb[u][q][j].w, acc[i][j], 0, 0, 0);
            }
        if (kt + 2 < nk) store_r(r2, u);        // tile kt+2 (loaded two steps ago into set u)
        if (kt + 4 < nk) load_g(kt + 4, u);     // four steps ahead, into the set just stored
        __syncthreads();
        ring = r1;
      }
    }
  } else {
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nk) load_global(u, u);
  store_lds(0, 0);
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {   // tile kt lives in register slot kt % PF == u (kt0 is a multiple of PF)
      const int kt = kt0 + u;
      if (kt >= nk) break;
      const int buf = kt & 1;
      if (kt + PF < nk && !(dbg & 2)) load_global(kt + PF, u);   // slot u is free: tile kt went to LDS one step ago
      const float* Ab = As + buf * BM * LS + (wr * (BM / WM) + frow) * LS + fk;
      const float* Wb = Ws + buf * BN * LS + (wc * (BN / WN) + frow) * LS + fk;
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
      if (kt + 1 < nk) store_lds(buf ^ 1, (u + 1) % PF);
      __syncthreads();
    }
  }
  }

  if (dbg & 1) {   // tuning aid: no epilogue at all (keeps the accumulators alive through one dummy store)
    if (acc[0][0][0] == 123456.789f) a.C[0] = acc[0][0][1];
    return;
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

template <int BM, int BN, int PRO, int EPI, int BK = 32, int PF = 1, int PIPE = 0>
static int launch_nt(const GemmArgs& a, hipStream_t st) {
  constexpr int LS = BK + 4;
  const long long nblk = 8LL * cdiv(cdiv(a.M, BM), 8) * cdiv(a.N, BN);
  if (nblk * 256 >= (1LL << 32)) return fail(UR_ERR_UNSUPPORTED, "gemm_nt: %lld workgroups exceed HIP's 2^32-thread grid limit", nblk);
  dim3 grid((unsigned)nblk);
  size_t lds = (size_t)(PIPE ? 3 : 2) * (BM + BN) * LS * sizeof(float);
  const size_t cs = (size_t)BM * (BN + 4) * sizeof(float);
  if (cs > lds) lds = cs;
  static const hipError_t attr = hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, PRO, EPI, BK, PF, PIPE>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  UR_LAUNCH_EV((gemm_nt_kernel<BM, BN, PRO, EPI, BK, PF, PIPE>), grid, dim3(256), lds, st, a);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// Few rows (the last-row layer, the GRU's per-step GEMM): 64-row tiles leave most CUs idle and each workgroup MFMA-bound
// on its own K loop (M = 512, N = 128, K = 512: 8 workgroups x 13.7 us of MFMA).  32-row tiles double the workgroups.
// software-pipelined K loop (gemm_nt_kernel PIPE = 1) for the many-row launches: UR_GEMM_PIPE=0 / 1
static bool pipe_on(const GemmArgs& a) {
  static const int v = getenv("UR_GEMM_PIPE") ? atoi(getenv("UR_GEMM_PIPE")) : 0;
  return v != 0 && a.M > 1024;
}

static bool small_m(const GemmArgs& a) {
  static const int force = getenv("UR_GEMM_SMALLM") ? atoi(getenv("UR_GEMM_SMALLM")) : -1;   // tuning aid: 0 = never, 1 = always
  if (force >= 0) return force == 1;
  return a.M <= 1024;
}

template <int PRO, int EPI>
static int dispatch_tile(const GemmArgs& a, hipStream_t st) {
  // enough 128x128 tiles to fill 256 CUs twice? otherwise use 64-row tiles for more workgroups
  const long long big = (long long)cdiv(a.M, 128) * cdiv(a.N, 128);
  static const int force = getenv("UR_GEMM_TILE") ? atoi(getenv("UR_GEMM_TILE")) : 0;   // tuning aid: 64 or 128 rows
  if (a.N <= 64) return launch_nt<64, 64, PRO, EPI>(a, st);
  if (small_m(a)) return launch_nt<32, 128, PRO, EPI>(a, st);
  // compacted rows and a one-tile-wide output: the 64-row grid (M/64 workgroups, ~1.5 per CU) hides the rows that were
  // skipped behind wave quantisation; 32-row tiles let the saving through
  // ... as 64 x 64 tiles (same workgroup count as 32 x 128, 16 KB instead of 20 KB of operands per K-step: measured 1 % of the
  // step); UR_GEMM_C64=0 restores the 32-row tiles
  static const int c64 = getenv("UR_GEMM_C64") ? atoi(getenv("UR_GEMM_C64")) : 1;   // tuning aid
  if (a.m_dev && a.N <= 128 && a.N > 64 && c64 == 1)
    return pipe_on(a) ? launch_nt<64, 64, PRO, EPI, 32, 1, 1>(a, st) : launch_nt<64, 64, PRO, EPI>(a, st);
  if (a.m_dev && a.N <= 128) return launch_nt<32, 128, PRO, EPI>(a, st);
  // short K, wide N (QKV, FFN-1, d-act): the 16-deep K-step variant keeps 4 workgroups per CU resident and measured
  // 6-10 % faster at M = 25600; elsewhere the 32-deep step wins
  if (force == 16 || (force == 0 && a.K <= 128 && a.N >= 256))
    return pipe_on(a) ? launch_nt<64, 128, PRO, EPI, 16, 1, 1>(a, st) : launch_nt<64, 128, PRO, EPI, 16>(a, st);
  if (force == 1616) return launch_nt<128, 128, PRO, EPI, 16>(a, st);
  if (force == 128 || (force == 0 && big >= 512)) return launch_nt<128, 128, PRO, EPI>(a, st);
  return launch_nt<64, 128, PRO, EPI>(a, st);
}

// Rows per tile of the launches whose epilogue needs whole rows (LayerNorm forward / backward: BN = 128 = one row).  UR_GEMM_LNTILE
// = 32 / 64 / 128 (tuning aid).  One 32 x 32 accumulator per wave (32-row tiles) leaves 16 dependent MFMAs between two barriers and
// twice the LDS bytes per MFMA of the 64 x 64 register block of the 128-row tile.
static int ln_tile_rows() {
  static const int v = getenv("UR_GEMM_LNTILE") ? atoi(getenv("UR_GEMM_LNTILE")) : 32;
  return v;
}
static int lnbwd_bm(int M) { return (M > 1024 && ln_tile_rows() != 32) ? ln_tile_rows() : 32; }
int gemm_nt_lnbwd_tiles(int M) { return cdiv(M, lnbwd_bm(M)); }

int gemm_nt(const GemmArgs& a, int pro, int epi, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return UR_OK;
  ProfScope ps(PC_GEMM_NT, st, 2.0 * a.M * a.N * a.K);
  static const int dbg_env = getenv("UR_GEMM_DEBUG") ? atoi(getenv("UR_GEMM_DEBUG")) : 0;
  const_cast<GemmArgs&>(a).debug = dbg_env;
  if ((a.K & 3) || (a.N & 3) || (a.lda & 3) || (a.ldw & 3) || (epi != EPI_COUNT_GT && ((a.ldc & 3) || (a.aux && (a.ldaux & 3)) || (a.aux2 && (a.ldaux2 & 3)))))
    return fail(UR_ERR_ARG, "gemm_nt: N, K and all leading dimensions must be multiples of 4 (N=%d K=%d)", a.N, a.K);
  if (epi == EPI_BIAS_RES_LN) {
    if (a.N > 256 || a.ldc != a.N) return fail(UR_ERR_UNSUPPORTED, "gemm_nt: fused LayerNorm needs N<=256 (N=%d)", a.N);
    if (a.N <= 128 && !small_m(a) && ln_tile_rows() == 128)
      return pro == PRO_ACT ? launch_nt<128, 128, PRO_ACT, EPI_BIAS_RES_LN>(a, st)
                            : launch_nt<128, 128, PRO_NONE, EPI_BIAS_RES_LN>(a, st);
    if (a.N <= 128 && !small_m(a) && ln_tile_rows() == 64)
      return pro == PRO_ACT ? launch_nt<64, 128, PRO_ACT, EPI_BIAS_RES_LN>(a, st)
                            : launch_nt<64, 128, PRO_NONE, EPI_BIAS_RES_LN>(a, st);
    if (a.N <= 128 && a.m_dev && pipe_on(a))
      return pro == PRO_ACT ? launch_nt<32, 128, PRO_ACT, EPI_BIAS_RES_LN, 32, 1, 1>(a, st)
                            : launch_nt<32, 128, PRO_NONE, EPI_BIAS_RES_LN, 32, 1, 1>(a, st);
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
    switch (lnbwd_bm(a.M)) {
      case 128: return launch_nt<128, 128, PRO_NONE, EPI_ADD_LNBWD>(a, st);
      case 64: return launch_nt<64, 128, PRO_NONE, EPI_ADD_LNBWD>(a, st);
      case 96: return launch_nt<96, 128, PRO_NONE, EPI_ADD_LNBWD>(a, st);
      default: return pipe_on(a) ? launch_nt<32, 128, PRO_NONE, EPI_ADD_LNBWD, 32, 1, 1>(a, st)
                                 : launch_nt<32, 128, PRO_NONE, EPI_ADD_LNBWD>(a, st);
    }
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
constexpr int TB = 128;  // output tile (rows of Out = columns of P) x (cols of Out = columns of Q)
constexpr int BT = 32;   // tokens per LDS stage

// NW = 4 waves (2 x 2, 64 x 64 per wave) or 8 waves (2 x 4, 64 x 32 per wave: with ONE workgroup per CU the second wave of
// every SIMD issues MFMAs while the first waits on LDS / the barrier)
template <int PRO, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_tn_kernel(const float* __restrict__ P, int ldp, const float* __restrict__ Q,
                                                      int ldq, int T, int R, int Cc, int tok_per_split, int n_splits, int act,
                                                      float* __restrict__ part, float* __restrict__ bias_part,
                                                      const int* __restrict__ t_dev, const float* __restrict__ zero_row, int swz) {
  if (t_dev) {   // compacted token rows: spread the ACTUAL tokens over the splits (the host sized the split for the maximum)
    T = min(T, *t_dev);
    tok_per_split = (((T + n_splits - 1) / n_splits + BT - 1) / BT) * BT;
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // a fragment read takes 32 consecutive floats of token t (lanes 0-31) and of token t + 1 (lanes 32-63): with the rows 128 floats apart the
  // two halves would hit the same 32 banks (2-way conflict on every read), so the rows of ODD tokens are stored with column bit 5 flipped
  // (c ^ 32): the halves then sit in different banks, and no LDS is added
  float* Ps = smem;                  // [2][BT*TB]
  float* Qs = smem + 2 * BT * TB;    // [2][BT*TB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int WC = NW / 2, TN_ = 4 / WC;   // waves along the columns, 32-column MFMA tiles per wave
  constexpr int NT_ = 64 * NW, RP = NT_ / 32, NP = BT / RP;   // threads, token rows per load pass, passes per stage
  const int wr = wave / WC, wc = wave % WC;
  // XCD-aware mapping (see gemm_nt): the output tiles of one token split share an XCD, so the P / Q rows of that
  // split are fetched into one L2 only
  const int ntc = (Cc + TB - 1) / TB, ntiles = ntc * ((R + TB - 1) / TB);
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
  const int sp = (qid / ntiles) * 8 + xcd, tile = qid % ntiles;
  if (sp >= n_splits) return;
  const int c0 = (tile % ntc) * TB, r0 = (tile / ntc) * TB;
  const bool first_ctile = (tile % ntc) == 0;
  const int t_begin = sp * tok_per_split;
  const int t_end = min(T, t_begin + tok_per_split);
  const int c4 = tid & 31, trow = tid >> 5;  // RP token rows per pass, NP passes

  floatx16 acc[2][TN_];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;

  const bool rin = r0 + c4 * 4 < R, cin = c0 + c4 * 4 < Cc;
  const float* Pp = P + (rin ? r0 + c4 * 4 : 0);
  const float* Qp = Q + (cin ? c0 + c4 * 4 : 0);
  typedef float tfx4 __attribute__((ext_vector_type(4)));   // (plain LLVM vectors: arrays of HIP's float4 class are not always kept in registers)
  tfx4 rp[NP], rq[NP];
  auto load_global = [&](int t0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      // rows past the range (and the column groups past R) load ZEROS from a spare row instead of being zeroed after the load: a select
      // on the loaded value makes the compiler branch around it and wait for the load on the spot (s_waitcnt vmcnt(0) at the TOP of the
      // stage: the prefetch of the next stage was not overlapping the MFMAs at all).  P = 0 makes the product 0 whatever Q holds.
      const int t = t0 + trow + RP * i;
      const bool tin = t < t_end;
      const int tt = min(t, T - 1);
      const float* pp = (tin && rin) ? Pp + (long long)t * ldp : zero_row;
      rp[i] = *(const tfx4*)pp;
      rq[i] = *(const tfx4*)(Qp + (long long)tt * ldq);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      *(tfx4*)(Ps + buf * BT * TB + (trow + RP * i) * TB + ((c4 * 4) ^ (((trow + RP * i) & swz) << 5))) = rp[i];
      float4 v = make_float4(rq[i].x, rq[i].y, rq[i].z, rq[i].w);
      if (PRO == PRO_ACT) v = act4(v, act);
      *(float4*)(Qs + buf * BT * TB + (trow + RP * i) * TB + ((c4 * 4) ^ (((trow + RP * i) & swz) << 5))) = v;
    }
  };

  const int nt = (t_end - t_begin + BT - 1) / BT;
  if (nt > 0) {
    load_global(t_begin);
    store_lds(0);
  }
  __syncthreads();
  const int fcol = lane & 31, ft = lane >> 5;
  const int sw = (ft & swz) << 5;   // column swizzle of this lane's (odd/even) token rows
  int qc[TN_];              // swizzled column offsets of this wave's Q fragments
#pragma unroll
  for (int j = 0; j < TN_; ++j) qc[j] = (wc * (32 * TN_) + 32 * j) ^ sw;
  for (int it = 0; it < nt; ++it) {
    const int buf = it & 1;
    if (it + 1 < nt) load_global(t_begin + (it + 1) * BT);
    const float* Pb = Ps + buf * BT * TB + ft * TB + wr * 64 + fcol;
    const float* Qb = Qs + buf * BT * TB + ft * TB + fcol;
    // fragments of step kk + FD are read while step kk's MFMAs run (written out: left to itself the compiler waits for each step's
    // reads right in front of that step's MFMAs, lgkmcnt(0) sixteen times per stage)
    constexpr int FD = 4;                      // steps of read-ahead
    float fa0[FD], fa1[FD], fbq[FD][TN_];
#pragma unroll
    for (int u = 0; u < FD; ++u) {
      fa0[u] = Pb[2 * u * TB + sw];
      fa1[u] = Pb[2 * u * TB + (sw ^ 32)];
#pragma unroll
      for (int j = 0; j < TN_; ++j) fbq[u][j] = Qb[2 * u * TB + qc[j]];
    }
#pragma unroll
    for (int kk = 0; kk < BT; kk += 2) {
      const int u = (kk >> 1) % FD;
      const float a0 = fa0[u], a1 = fa1[u];
      float b[TN_];
#pragma unroll
      for (int j = 0; j < TN_; ++j) b[j] = fbq[u][j];
      if (kk + 2 * FD < BT) {
        fa0[u] = Pb[(kk + 2 * FD) * TB + sw];
        fa1[u] = Pb[(kk + 2 * FD) * TB + (sw ^ 32)];
#pragma unroll
        for (int j = 0; j < TN_; ++j) fbq[u][j] = Qb[(kk + 2 * FD) * TB + qc[j]];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[j], acc[1][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bias_part != nullptr && first_ctile && tid < TB) {
#pragma unroll 8
      for (int t = 0; t < BT; ++t) bsum += Ps[buf * BT * TB + t * TB + (tid ^ ((t & swz) << 5))];
    }
    if (it + 1 < nt) store_lds(buf ^ 1);
    __syncthreads();
  }

  // epilogue through LDS -> coalesced float4 row stores of the partial tile
  constexpr int CS = TB + 4;
  float* Cs = smem;  // [TB][CS] = 67.6 KB (launch reserves max(staging, this))
  {
    const int lcol = lane & 31, lrow4 = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN_; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[(wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + lrow4) * CS + wc * (32 * TN_) + j * 32 + lcol] = acc[i][j][r];
  }
  __syncthreads();
  float* out = part + (long long)sp * R * Cc;
  {
    const int t = tid & 31, g = tid >> 5;
    const int c = c0 + t * 4;
    if (c < Cc)
      for (int rl = g; rl < TB; rl += RP) {
        const int rr = r0 + rl;
        if (rr >= R) break;
        *(float4*)(out + (long long)rr * Cc + c) = *(const float4*)(Cs + rl * CS + t * 4);
      }
  }
  if (bias_part != nullptr && first_ctile && tid < TB && r0 + tid < R) bias_part[(long long)sp * R + r0 + tid] = bsum;
}

// ---- the same product with NO LDS: operand fragments straight from global memory.
// For Out = P^T Q both operands are walked along the REDUCTION dimension row by row, and a 32x32x2 MFMA fragment is exactly that: lane
// (i, k) of the A operand holds P[t + k][r + i] -- 32 consecutive floats of one token row per half-wave, a full 128-byte line.  So a
// wave loads its fragments with plain coalesced global loads and the workgroup shares nothing: no staging stores, no barriers, no LDS
// allocation (the staged kernel's 67.6 KB decide on which CUs it can run next to the main stream's kernels), and the waves of a CU
// drift freely.  float2 loads take TWO adjacent columns per lane (columns 2i, 2i+1 of a 64-column block -> two MFMAs; the output
// tile comes out with its rows / columns interleaved the same way, undone by the addressing of the epilogue), so an iteration (two
// tokens, four MFMAs = 256 MFMA cycles per wave) costs two load instructions.  Latency is covered by a register ring: the loads of
// iteration it + DEPTH are issued when iteration it is consumed (DEPTH = 16: 2 x 16 float2 = 64 VGPRs, ~4 000 MFMA cycles ahead).
// Redundant fetches (each P element is wanted by the two waves of a row pair, each Q element by two waves of a column pair, and by
// the other column / row tiles of the split) are served by L1 / the XCD's L2: the split -> XCD mapping is the staged kernel's.
// Workgroup = 4 waves (2 x 2), 128 x 128 output tile, 64 x 64 per wave (four accumulators: consecutive MFMAs never depend).
template <int PRO, int DEPTH>
__global__ __launch_bounds__(256) void gemm_tn_direct_kernel(const float* __restrict__ P, int ldp, const float* __restrict__ Q, int ldq,
                                                             int T, int R, int Cc, int tok_per_split, int n_splits, int act,
                                                             float* __restrict__ part, float* __restrict__ bias_part,
                                                             const int* __restrict__ t_dev, const float* __restrict__ tn_zero) {
  if (t_dev) {
    T = min(T, *t_dev);
    tok_per_split = (((T + n_splits - 1) / n_splits + BT - 1) / BT) * BT;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kk = lane >> 5;
  const int ntc = (Cc + TB - 1) / TB, ntiles = ntc * ((R + TB - 1) / TB);
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
  const int sp = (qid / ntiles) * 8 + xcd, tile = qid % ntiles;
  if (sp >= n_splits) return;
  const int c0 = (tile % ntc) * TB, r0 = (tile / ntc) * TB;
  const bool first_ctile = (tile % ntc) == 0;
  const int t_begin = sp * tok_per_split;
  const int t_end = min(T, t_begin + tok_per_split);
  const int ra = r0 + wr * 64 + 2 * li, cb = c0 + wc * 64 + 2 * li;   // this lane's two P columns (= output rows) / Q columns
  const bool rin = ra < R, cin = cb < Cc;
  const float* Pp = P + (rin ? ra : 0);
  const float* Qp = Q + (cin ? cb : 0);

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float2 bsum = make_float2(0.f, 0.f);

  const int n_it = (max(t_end - t_begin, 0) + 1) >> 1;
  // TWO rings in ping-pong: a round consumes one and refills the other (DEPTH iterations ahead).  With a single ring refilled in place
  // the old value of a slot is still wanted by its MFMAs when the new load is issued, the register allocator gives the load another
  // register and rotates the whole ring at the back-edge (32 v_movs behind an s_waitcnt vmcnt(0): a full drain every round); here
  // every slot has one definition per round trip and no copy.  sched_barrier: the loads stay where they are written (the scheduler
  // would sink them to just ahead of their use).
  typedef float fx2 __attribute__((ext_vector_type(2)));
  fx2 fa[DEPTH], fb[DEPTH], ga[DEPTH], gb[DEPTH];
  // out-of-range tokens (and the lanes of a column pair past R) read a ZERO instead of being zeroed after the load: a select on the
  // loaded value lets the optimizer sink the load into the branch that uses it (a dependent round trip per iteration)
  auto issue = [&](int it, fx2& a, fx2& b) {
    const int t = t_begin + 2 * it + kk;
    const bool in = rin && t < t_end;
    const float* pa = in ? Pp + (long long)t * ldp : tn_zero;
    a = *(const fx2*)pa;
    b = *(const fx2*)(Qp + (long long)min(t, T - 1) * ldq);
  };
  auto consume = [&](int it, fx2 a, fx2 b) {
    if (PRO == PRO_ACT) b = fx2{act_fwd(b.x, act), act_fwd(b.y, act)};
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
    bsum.x += a.x;
    bsum.y += a.y;
  };
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) issue(u, fa[u], fb[u]);
  for (int base = 0; base < n_it; base += 2 * DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      issue(base + DEPTH + u, ga[u], gb[u]);
      __builtin_amdgcn_sched_barrier(0);
      consume(base + u, fa[u], fb[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      issue(base + 2 * DEPTH + u, fa[u], fb[u]);
      __builtin_amdgcn_sched_barrier(0);
      consume(base + DEPTH + u, ga[u], gb[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // accumulator element r of lane (li, kk): fragment row (r & 3) + 8 (r >> 2) + 4 kk, fragment column li; fragment row / column f of
  // accumulator [i][j] is output row 2 f + i / output column 2 f + j of the wave's 64 x 64 block
  float* out = part + (long long)sp * R * Cc;
  if (cin) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = r0 + wr * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * kk) + i;
        if (rr < R) *(float2*)(out + (long long)rr * Cc + cb) = make_float2(acc[i][0][r], acc[i][1][r]);
      }
  }
  if (bias_part != nullptr && first_ctile && wc == 0) {
    bsum.x += __shfl_xor(bsum.x, 32, 64);
    bsum.y += __shfl_xor(bsum.y, 32, 64);
    if (kk == 0 && rin) *(float2*)(bias_part + (long long)sp * R + ra) = bsum;
  }
}

// out[i] = sum_s part[s*n + i] (fixed order; 4 consecutive elements per thread); the same launch also reduces the
// bias partials bias_part[s*R + r] -> bias_out[r] (threads past n/4)
__global__ void reduce_splits_kernel(const float* __restrict__ part, int S, long long n, int cols, float* __restrict__ out,
                                     int ldo, const float* __restrict__ bias_part, int R, float* __restrict__ bias_out) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nq = n / 4;
  if (q < nq) {
    const long long i = q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int s = 0; s < S; ++s) {
      const float4 v = *(const float4*)(part + (long long)s * n + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *(float4*)(out + (i / cols) * ldo + (i % cols)) = acc;   // cols % 4 == 0 keeps the 4 elements in one row
  } else if (bias_out != nullptr && q - nq < R / 4) {
    const long long i = (q - nq) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int s = 0; s < S; ++s) {
      const float4 v = *(const float4*)(bias_part + (long long)s * R + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *(float4*)(bias_out + i) = acc;
  }
}

// a few zero floats in device memory (per device, allocated once): what the direct kernel's out-of-range lanes load
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

static int tn_splits(int T, int R, int Cc) {
  const int tiles = cdiv(R, TB) * cdiv(Cc, TB);
  // one workgroup per CU: every further split is another R x Cc partial tile written and read back by the reduction, which
  // shares HBM with the sparse optimizer step (measured at the C5 shapes: 256 -> 0.872 ms/step, 512 -> 0.882, 128 -> 0.94)
  static const int target = getenv("UR_TN_BLOCKS") ? atoi(getenv("UR_TN_BLOCKS")) : 256;   // tuning aid
  // Never more workgroups on an XCD than it has CUs: workgroup number 33 of an XCD shares a CU with another one, both run at half
  // speed and the launch takes twice as long.  Split sp runs on XCD sp % 8 (all its tiles together), so the bound is
  // ceil(S / 8) * tiles <= target / 8:  3 tiles x 86 splits (11 splits = 33 workgroups on five of the XCDs) took 45 us, 3 x 80 takes 27.
  int s = (target / 8) / tiles * 8;
  if (s < 8) s = target / tiles;
  const int smax = cdiv(T, T <= 4096 ? 64 : 128);   // >= 2 (small T) / 4 LDS stages of 32 tokens per split
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  return s;
}

long long gemm_tn_group_ws_floats(int R, int Cc);
static bool tn_use_group() {   // UR_TN_GROUP=0: the single-product kernel (round 1-2) for every weight-gradient GEMM
  static const bool v = !(getenv("UR_TN_GROUP") && atoi(getenv("UR_TN_GROUP")) == 0);
  return v;
}
bool gemm_tn_grouped() { return tn_use_group(); }
long long gemm_tn_ws_floats(int T, int R, int Cc) {
  const int s = tn_splits(T, R, Cc);
  return std::max((long long)s * R * Cc + (long long)s * R + 64, gemm_tn_group_ws_floats(R, Cc));
}

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

int reduce_batch(ReduceBatch& rb, hipStream_t st) {
  if (rb.n <= 0) return UR_OK;
  ProfScope ps(PC_GEMM_TN, st, 0.0);
  int blocks = 0;
  for (int i = 0; i < rb.n; ++i) {
    rb.item[i].first_block = blocks;
    blocks += cdiv(rb.item[i].n / 4, 16);
  }
  hipLaunchKernelGGL(reduce_batch_kernel, dim3(blocks), dim3(256), 0, st, rb);
  UR_LAUNCH_CHECK();
  rb.n = 0;
  return UR_OK;
}

int gemm_tn(const float* P, int ldp, const float* Q, int ldq, int T, int R, int Cc, int pro_act_on_q, int act, float* out,
            int ldo, float* bias_out, float* ws, hipStream_t st, ReduceBatch* defer, const int* t_dev) {
  if ((R & 3) || (Cc & 3) || (ldp & 3) || (ldq & 3) || (ldo & 3)) return fail(UR_ERR_ARG, "gemm_tn: R/Cc/ld must be multiples of 4");
  if (T <= 0) return fail(UR_ERR_ARG, "gemm_tn: T=%d", T);
  if (tn_use_group()) {
    const TnReq q{P, ldp, Q, ldq, T, R, Cc, pro_act_on_q, act, out, ldo, bias_out, ws, t_dev};
    return gemm_tn_group(&q, 1, st, defer);
  }
  ProfScope ps(PC_GEMM_TN, st, 2.0 * T * R * Cc);
  const int S = tn_splits(T, R, Cc);
  int tps = cdiv(T, S);
  tps = cdiv(tps, BT) * BT;
  float* part = ws;
  float* bias_part = bias_out ? ws + (long long)S * R * Cc : nullptr;
  dim3 grid(8 * cdiv(S, 8) * cdiv(Cc, TB) * cdiv(R, TB));
  const size_t lds = (size_t)TB * (TB + 4) * sizeof(float);  // >= 4*BT*TB staging
  static const int nw = getenv("UR_TN_WAVES") ? atoi(getenv("UR_TN_WAVES")) : 8;   // tuning aid: 4 or 8 waves per workgroup
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<PRO_ACT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<PRO_NONE, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<PRO_ACT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<PRO_NONE, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  static const int swz = (getenv("UR_TN_NO_SWIZZLE") && atoi(getenv("UR_TN_NO_SWIZZLE"))) ? 0 : 1;   // 1: odd token rows stored with column bit 5 flipped (no bank conflicts)
  const float* zeros_staged = tn_zero_buf();
  if (!zeros_staged) return fail(UR_ERR_HIP, "gemm_tn: no device memory for the zero row");
  static const int direct = getenv("UR_TN_DIRECT") ? atoi(getenv("UR_TN_DIRECT")) : 0;   // 0 (default): the LDS-staged kernel; 8 / 16: the no-LDS kernel, ring depth
  if (direct) {
    const float* zeros = tn_zero_buf();
    if (!zeros) return fail(UR_ERR_HIP, "gemm_tn: no device memory for the zero row");
#define UR_TND_GO(PRO_, DP_) hipLaunchKernelGGL((gemm_tn_direct_kernel<PRO_, DP_>), grid, dim3(256), 0, st, P, ldp, Q, ldq, T, R, Cc, tps, S, act, part, bias_part, t_dev, zeros)
    if (direct == 16) { if (pro_act_on_q) UR_TND_GO(PRO_ACT, 16); else UR_TND_GO(PRO_NONE, 16); }
    else { if (pro_act_on_q) UR_TND_GO(PRO_ACT, 8); else UR_TND_GO(PRO_NONE, 8); }
#undef UR_TND_GO
  } else
#define UR_TN_GO(PRO_, NW_) hipLaunchKernelGGL((gemm_tn_kernel<PRO_, NW_>), grid, dim3(64 * NW_), lds, st, P, ldp, Q, ldq, T, R, Cc, tps, S, act, part, bias_part, t_dev, zeros_staged, swz)
  if (nw == 4) { if (pro_act_on_q) UR_TN_GO(PRO_ACT, 4); else UR_TN_GO(PRO_NONE, 4); }
  else { if (pro_act_on_q) UR_TN_GO(PRO_ACT, 8); else UR_TN_GO(PRO_NONE, 8); }
#undef UR_TN_GO
  UR_LAUNCH_CHECK();
  const long long n = (long long)R * Cc;
  if (defer) {
    if (defer->full(2)) {
      int rc = reduce_batch(*defer, st);
      if (rc) return rc;
    }
    defer->add(part, n, S, n, Cc, out, ldo);
    if (bias_out) defer->add(bias_part, R, S, R, R, bias_out, R);
    return UR_OK;
  }
  hipLaunchKernelGGL(reduce_splits_kernel, dim3(cdiv(n / 4 + (bias_out ? R / 4 : 0), 256)), dim3(256), 0, st, part, S, n, Cc, out, ldo,
                     bias_part, R, bias_out);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// ======================================================================================== gemm_tn_group
// Several weight-gradient products Out_i[R_i, C_i] = P_i[T_i, R_i]^T pro(Q_i)[T_i, C_i] in ONE launch (round 3).  The single-product
// kernel above gives every GEMM its own launch and fills the chip by splitting the token dimension ~86 ways: 9-11 launches per backward
// pass, each workgroup runs 8 LDS stages between a cold prologue and a 64 KB partial-tile store, and the deferred reduction then reads
// 85-100 MB of partial tiles back (1.77 x the algorithmic HBM bytes; 0.17 of the fp32-MFMA roof in situ).  Here the chip is filled
// ACROSS products: every (product, 64 x 64 output tile, token split) is one workgroup of a single grid, so a layer's dW_2, dW_1, dW_o
// (36 tiles) need only 8 token splits -- one per XCD -- for 288 workgroups, each walking ~2 700 tokens (83 stages) per prologue /
// epilogue; the partial tiles shrink to S x R x C = a few MB per launch, and a product whose T is small (the B last rows) takes S = 1
// and writes its result (and bias gradient) directly.  The second stage of the split is the deferred ReduceBatch as before (fixed
// order: bit-reproducible).
//   workgroup = 4 waves (2 x 2), wave = one 32 x 32 accumulator; stage = 32 tokens x (64 + 64) columns, double buffered (32 KB);
//   odd token rows are stored with column bit 5 flipped (the two half-waves of a fragment read hit different banks);
//   out-of-range token rows load a zero row (pointer select, see gemm_tn_kernel); the bias gradient colsum(P) is accumulated from the
//   staging registers (no LDS reads) by the workgroups of tile column 0.
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
struct TnGroup { static constexpr int MAX = 12; TnItem item[MAX]; int n; };

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

constexpr int TN_GROUP_SMAX = 32;
long long gemm_tn_group_ws_floats(int R, int Cc) { return (long long)TN_GROUP_SMAX * ((long long)R * Cc + R) + 64; }

// req[i].ws: gemm_tn_group_ws_floats(R, Cc) floats each (untouched until the deferred reduction has run).  defer == nullptr: the
// reduction of the split products runs right behind the launch.
int gemm_tn_group(const TnReq* req, int n, hipStream_t st, ReduceBatch* defer) {
  if (n <= 0) return UR_OK;
  if (n > TnGroup::MAX) return fail(UR_ERR_ARG, "gemm_tn_group: %d products (max %d)", n, TnGroup::MAX);
  const float* zeros = tn_zero_buf();
  if (!zeros) return fail(UR_ERR_HIP, "gemm_tn: no device memory for the zero row");
  static const int target = getenv("UR_TN_BLOCKS") ? atoi(getenv("UR_TN_BLOCKS")) : 864;   // tuning aid: workgroups per launch
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
  ProfScope ps(PC_GEMM_TN, st, flops);
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
  // (tuning aid, UR_TN_LDS_KB: workgroup LDS footprint in KB -- reserved, unused bytes beyond the kernel's own 32 KB decide how many of
  // its workgroups fit on a CU next to the main stream's kernels)
  static const int lds_kb = getenv("UR_TN_LDS_KB") ? atoi(getenv("UR_TN_LDS_KB")) : 0;
  const size_t extra = lds_kb > 32 ? (size_t)(lds_kb - 32) * 1024 : 0;
  static bool attr_set = false;
  if (extra && !attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)extra);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_group_kernel, dim3(blocks), dim3(256), extra, st, g, zeros);
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
  int j = 0;
  while (j + 1 < tb.n && (int)blockIdx.x >= tb.item[j + 1].first_block) ++j;
  const TransposeItem it = tb.item[j];
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
  if (tb.n <= 0) return UR_OK;
  ProfScope ps(PC_MISC, st, 0.0);
  int blocks = 0;
  for (int i = 0; i < tb.n; ++i) {
    tb.item[i].first_block = blocks;
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
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(blocks), dim3(256), 0, st, tb);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

}  // namespace ur

// ---- raw entry points (unit tests and micro-benchmarks of the GEMM kernels; see include/unirec_amd.h)
extern "C" int ur_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, int pro,
                          int epi, int act, const float* bias, const float* aux, int ldaux, const float* gamma, const float* beta,
                          float eps, float* xhat, float* rstd, void* stream) {
  UR_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, UR_ERR_ARG, "ur_gemm_nt: bad argument");
  ur::GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.act = act; g.bias = bias;
  g.aux = aux; g.ldaux = ldaux; g.gamma = gamma; g.beta = beta; g.eps = eps; g.xhat = xhat; g.rstd = rstd;
  return ur::gemm_nt(g, pro, epi, ur::as_stream(stream));
}
extern "C" int64_t ur_gemm_tn_workspace_floats(int T, int R, int Cc) { return ur::gemm_tn_ws_floats(T, R, Cc); }
extern "C" int ur_gemm_tn(const float* P, int ldp, const float* Q, int ldq, int T, int R, int Cc, int pro_act_on_q, int act,
                          float* out, int ldo, float* bias_out, float* ws, void* stream) {
  UR_REQUIRE(P && Q && out && ws, UR_ERR_ARG, "ur_gemm_tn: null pointer");
  return ur::gemm_tn(P, ldp, Q, ldq, T, R, Cc, pro_act_on_q, act, out, ldo, bias_out, ws, ur::as_stream(stream));
}
