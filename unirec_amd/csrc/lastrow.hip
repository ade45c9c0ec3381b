// The last-row layer of the SASRec encoder as TWO launches (round 4).  sasrec.py:75 reads output[:, -1] only, so in the top layer
// everything behind the K / V projection is needed for ONE row per sequence: B rows instead of B * L.  Rounds 2-3 ran those B rows
// through the many-row kernels' decomposition (32-row blocks, 32x32x2 MFMA tiles, the inner dimension split over workgroups that
// exchange partial tiles through device-scope atomics): nine launches, 125 us, every one a latency chain on a mostly idle chip.
//
//   lastrow_fwd : x_L -> q = x_L Wq^T + bq -> one-query attention over the sequence's K / V rows -> a = LN(ctx Wo^T + bo + x_L)
//                 -> h1 = a W1^T + b1 -> y = LN(act(h1) W2^T + b2 + a)                 (modules.py:284-316, 347-355 for row L-1)
//   lastrow_bwd : the mirror image down to dq, dK, dV, and then the layer's WHOLE input gradient: with one query per (sequence, head)
//                 dK_j = ds[h,j] q_h and dV_j = p[h,j] dctx_h are rank one per head, so [dK dV]_j Wkv = sum_h ds[h,j] (q_h Wk_h) +
//                 p[h,j] (dctx_h Wv_h): a [len, 2H] x [2H, d] product per sequence (K = 32 instead of 256) -- the M-row projection-
//                 gradient GEMM (21 us) and the scatter-add GEMM behind it (15 us) disappear into this launch.
//
// Decomposition: R = 4 (or 8) sequences per workgroup, B / R workgroups of 8 waves, v_mfma_f32_4x4x1_16b_f32: sixteen independent
// 4 x 4 blocks per instruction at the full fp32 MFMA rate, the A operand only FOUR rows tall (the four rows = four sequences, lane l
// of block l / 4 supplies row l % 4), the 64 B lanes = 64 different output columns.  Weights are streamed K-MAJOR (Wt [K][N]: the
// transposed copies in the forward pass, the nn.Linear weights as stored in the backward pass), a lane owning VW = 2 or 4
// CONSECUTIVE columns: one fully coalesced buffer load per k-row and wave (512 B / 1 KB), VW MFMAs per load, straight into
// registers -- a workgroup pulls its 640-770 KB of weights at the texture path's full rate (tools/probe/wstream_probe.hip:
// 115 GB/s per workgroup against 35 for the lane-per-row pattern).  A phase's K range is split over the waves (N = d products: 8 K-parts)
// and the partial tiles are summed through LDS by the epilogue threads; every phase's weight fragments are requested before the
// previous phase's epilogue.  No cross-workgroup exchange, no counters, no atomics.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace ur {

typedef float fx4 __attribute__((ext_vector_type(4)));
typedef float fx2 __attribute__((ext_vector_type(2)));

constexpr int LR_NW = 8;          // waves per workgroup
constexpr int LR_THREADS = LR_NW * 64;
constexpr int LR_MAXL = 64;       // keys per sequence (= attn_compact_supported's bound)
constexpr int LR_K2MAX = 64;      // k-rows per wave of the K = inner products: inner <= 512

template <int VW> struct LrVec;
template <> struct LrVec<1> { typedef float T; };
template <> struct LrVec<2> { typedef fx2 T; };
template <> struct LrVec<4> { typedef fx4 T; };
template <int VW>
__device__ __forceinline__ float lr_comp(const typename LrVec<VW>::T& v, int c) {
  if constexpr (VW == 1) return v;
  else return v[c];
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lr_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}
template <int VW>
__device__ __forceinline__ typename LrVec<VW>::T lr_bload(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  if constexpr (VW == 4) return __builtin_bit_cast(fx4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
  else if constexpr (VW == 2) return __builtin_bit_cast(fx2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, soff, 0));
  else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, soff, 0));
}

// KN k-rows of a K-major weight matrix Wt (row stride ldt floats), columns col0 + VW * lane .. + VW - 1, rows k0 .. k0 + KN - 1: one
// coalesced load per row.  Everything wave-uniform goes into the scalar offset.
template <int VW, int KN>
__device__ __forceinline__ void lr_fetch(typename LrVec<VW>::T (&w)[KN], const float* Wt, int ldt, int col0, int k0, int lane) {
  const __amdgpu_buffer_rsrc_t rs = lr_rsrc(Wt);
  const unsigned voff = (unsigned)(lane * VW) * 4u;
  const int s0 = (k0 * ldt + col0) * 4, sl = ldt * 4;
#pragma unroll
  for (int k = 0; k < KN; ++k) w[k] = lr_bload<VW>(rs, voff, s0 + k * sl);
}

// acc[g][c] += A[4 g + lane % 4][k0 + k] * w[k][c]   (k < KN, g < RG row groups, c < VW column slots; A: LDS, row stride AS floats)
template <int VW, int KN, int RG>
__device__ __forceinline__ void lr_mma(fx4 (&acc)[RG][VW], const typename LrVec<VW>::T (&w)[KN], const float* A, int AS, int k0, int lane) {
  static_assert(KN % 4 == 0, "the A operand is read four k at a time");
  const float* ap = A + (lane & 3) * AS + k0;
#pragma unroll
  for (int k = 0; k < KN; k += 4) {
    float4 a4[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) a4[g] = *(const float4*)(ap + 4 * g * AS + k);
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int c = 0; c < VW; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].x, lr_comp<VW>(w[k], c), acc[g][c], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int c = 0; c < VW; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].y, lr_comp<VW>(w[k + 1], c), acc[g][c], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int c = 0; c < VW; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].z, lr_comp<VW>(w[k + 2], c), acc[g][c], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int c = 0; c < VW; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].w, lr_comp<VW>(w[k + 3], c), acc[g][c], 0, 0, 0);
  }
}
// the same with a run-time number of k-rows kn <= KMAX (kn % 4 == 0): register arrays are sized for KMAX, rows >= kn are neither
// loaded nor multiplied
template <int VW, int KMAX>
__device__ __forceinline__ void lr_fetch_rt(typename LrVec<VW>::T (&w)[KMAX], const float* Wt, int ldt, int col0, int k0, int kn, int lane) {
  const __amdgpu_buffer_rsrc_t rs = lr_rsrc(Wt);
  const unsigned voff = (unsigned)(lane * VW) * 4u;
  const int s0 = (k0 * ldt + col0) * 4, sl = ldt * 4;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < kn) w[k] = lr_bload<VW>(rs, voff, s0 + k * sl);
}
template <int VW, int KMAX, int RG>
__device__ __forceinline__ void lr_mma_rt(fx4 (&acc)[RG][VW], const typename LrVec<VW>::T (&w)[KMAX], const float* A, int AS, int k0, int kn, int lane) {
  const float* ap = A + (lane & 3) * AS + k0;
#pragma unroll
  for (int k = 0; k < KMAX; k += 4) {
    if (k < kn) {
      float4 a4[RG];
#pragma unroll
      for (int g = 0; g < RG; ++g) a4[g] = *(const float4*)(ap + 4 * g * AS + k);
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int c = 0; c < VW; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].x, lr_comp<VW>(w[k], c), acc[g][c], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int c = 0; c < VW; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].y, lr_comp<VW>(w[k + 1], c), acc[g][c], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int c = 0; c < VW; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].z, lr_comp<VW>(w[k + 2], c), acc[g][c], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int c = 0; c < VW; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].w, lr_comp<VW>(w[k + 3], c), acc[g][c], 0, 0, 0);
    }
  }
}
template <int VW, int RG>
__device__ __forceinline__ void lr_zero(fx4 (&acc)[RG][VW]) {
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int c = 0; c < VW; ++c) acc[g][c] = fx4{0.f, 0.f, 0.f, 0.f};
}
// accumulators -> out[(row0 + 4 g + v) * NS + col0 + VW * lane + c]   (register v of a 4x4x1 block = row v; lane = column)
template <int VW, int RG>
__device__ __forceinline__ void lr_put(const fx4 (&acc)[RG][VW], float* out, int NS, int row0, int col0, int lane) {
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float* p = out + (long long)(row0 + 4 * g + v) * NS + col0 + VW * lane;
      if constexpr (VW == 4) *(float4*)p = make_float4(acc[g][0][v], acc[g][1][v], acc[g][2][v], acc[g][3][v]);
      else if constexpr (VW == 2) *(float2*)p = make_float2(acc[g][0][v], acc[g][1][v]);
      else *p = acc[g][0][v];
    }
}

// ---- N = D products with 16-byte loads.  D / 4 lanes span a k-row of a K-major [K][D] matrix, so ONE wave instruction covers Q = 256 / D
// k-rows (D = 128: the two half-waves; D = 64: four quarter-waves): 1 KB per instruction, the texture path's full rate -- with one
// 8-byte load per lane (a 64-lane row of 128 columns) the same stream ran at 65 GB/s per workgroup against 115 (wstream_probe).
// Sub-row q = lane / (D / 4) walks ITS contiguous share of the wave's K range, k0 + q * (kn / Q) + s, and is a K-part of its own:
// LR_NW * Q partial tiles, part index wave * Q + q (the A operand of a 4x4x1 block comes from the block's own four lanes, which share q).
template <int D> struct LrH { static constexpr int LPR = D / 4, Q = 64 / LPR, NP = LR_NW * Q; };
template <int D, int SMAX>
__device__ __forceinline__ void lr_fetch_h(fx4 (&w)[SMAX], const float* Wt, int ldt, int k0, int kn, int lane) {
  const __amdgpu_buffer_rsrc_t rs = lr_rsrc(Wt);
  const int ks = kn / LrH<D>::Q;
  const unsigned voff = (unsigned)((lane / LrH<D>::LPR) * ks * ldt + 4 * (lane % LrH<D>::LPR)) * 4u;
  const int s0 = k0 * ldt * 4, sl = ldt * 4;
#pragma unroll
  for (int k = 0; k < SMAX; ++k)
    if (k < ks) w[k] = lr_bload<4>(rs, voff, s0 + k * sl);
}
template <int D, int SMAX, int RG>
__device__ __forceinline__ void lr_mma_h(fx4 (&acc)[RG][4], const fx4 (&w)[SMAX], const float* A, int AS, int k0, int kn, int lane) {
  const int ks = kn / LrH<D>::Q;
  const float* ap = A + (lane & 3) * AS + k0 + (lane / LrH<D>::LPR) * ks;
  if constexpr (SMAX % 4 == 0) {
    if ((ks & 3) == 0) {   // (wave-uniform) the A operand four k at a time
#pragma unroll
      for (int k = 0; k < SMAX; k += 4) {
        if (k < ks) {
          float4 a4[RG];
#pragma unroll
          for (int g = 0; g < RG; ++g) a4[g] = *(const float4*)(ap + 4 * g * AS + k);
#pragma unroll
          for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].x, w[k][c], acc[g][c], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].y, w[k + 1][c], acc[g][c], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].z, w[k + 2][c], acc[g][c], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g].w, w[k + 3][c], acc[g][c], 0, 0, 0);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < SMAX; ++k) {
    if (k < ks) {
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        const float av = ap[4 * g * AS + k];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, w[k][c], acc[g][c], 0, 0, 0);
      }
    }
  }
}
// partial tiles -> out[((wave * Q + q) * R + 4 g + v) * D + 4 * (lane % LPR) + c];  FOLD: the Q sub-rows of the wave are summed with
// shuffles first and sub-row 0 writes part `wave` (LR_NW parts)
template <int D, int RG, bool FOLD>
__device__ __forceinline__ void lr_put_h(fx4 (&acc)[RG][4], float* out, int wave, int lane) {
  constexpr int R = 4 * RG, LPR = LrH<D>::LPR, Q = LrH<D>::Q;
  const int q = lane / LPR, c4 = lane % LPR;
  if constexpr (FOLD) {
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float x = acc[g][c][v];
#pragma unroll
          for (int m = LPR; m < 64; m <<= 1) x += __shfl_xor(x, m, 64);
          acc[g][c][v] = x;
        }
    if (q != 0) return;
  }
  const int p = FOLD ? wave : wave * Q + q;
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      *(float4*)(out + (long long)(p * R + 4 * g + v) * D + 4 * c4) = make_float4(acc[g][0][v], acc[g][1][v], acc[g][2][v], acc[g][3][v]);
}

__device__ __forceinline__ float4 lr_act_fwd4(float4 v, int act) {
  switch (act) {   // ONE switch around the four evaluations (rowchain.hip: rc_act_fwd16)
    case UR_ACT_GELU: v.x = act_fwd(v.x, UR_ACT_GELU); v.y = act_fwd(v.y, UR_ACT_GELU); v.z = act_fwd(v.z, UR_ACT_GELU); v.w = act_fwd(v.w, UR_ACT_GELU); break;
    case UR_ACT_RELU: v.x = act_fwd(v.x, UR_ACT_RELU); v.y = act_fwd(v.y, UR_ACT_RELU); v.z = act_fwd(v.z, UR_ACT_RELU); v.w = act_fwd(v.w, UR_ACT_RELU); break;
    case UR_ACT_SWISH: v.x = act_fwd(v.x, UR_ACT_SWISH); v.y = act_fwd(v.y, UR_ACT_SWISH); v.z = act_fwd(v.z, UR_ACT_SWISH); v.w = act_fwd(v.w, UR_ACT_SWISH); break;
    case UR_ACT_TANH: v.x = act_fwd(v.x, UR_ACT_TANH); v.y = act_fwd(v.y, UR_ACT_TANH); v.z = act_fwd(v.z, UR_ACT_TANH); v.w = act_fwd(v.w, UR_ACT_TANH); break;
    case UR_ACT_SIGMOID: v.x = act_fwd(v.x, UR_ACT_SIGMOID); v.y = act_fwd(v.y, UR_ACT_SIGMOID); v.z = act_fwd(v.z, UR_ACT_SIGMOID); v.w = act_fwd(v.w, UR_ACT_SIGMOID); break;
    default: break;
  }
  return v;
}
__device__ __forceinline__ float4 lr_act_bwd4(float4 v, int act) {
  switch (act) {
    case UR_ACT_GELU: v.x = act_bwd(v.x, UR_ACT_GELU); v.y = act_bwd(v.y, UR_ACT_GELU); v.z = act_bwd(v.z, UR_ACT_GELU); v.w = act_bwd(v.w, UR_ACT_GELU); break;
    case UR_ACT_RELU: v.x = act_bwd(v.x, UR_ACT_RELU); v.y = act_bwd(v.y, UR_ACT_RELU); v.z = act_bwd(v.z, UR_ACT_RELU); v.w = act_bwd(v.w, UR_ACT_RELU); break;
    case UR_ACT_SWISH: v.x = act_bwd(v.x, UR_ACT_SWISH); v.y = act_bwd(v.y, UR_ACT_SWISH); v.z = act_bwd(v.z, UR_ACT_SWISH); v.w = act_bwd(v.w, UR_ACT_SWISH); break;
    case UR_ACT_TANH: v.x = act_bwd(v.x, UR_ACT_TANH); v.y = act_bwd(v.y, UR_ACT_TANH); v.z = act_bwd(v.z, UR_ACT_TANH); v.w = act_bwd(v.w, UR_ACT_TANH); break;
    case UR_ACT_SIGMOID: v.x = act_bwd(v.x, UR_ACT_SIGMOID); v.y = act_bwd(v.y, UR_ACT_SIGMOID); v.z = act_bwd(v.z, UR_ACT_SIGMOID); v.w = act_bwd(v.w, UR_ACT_SIGMOID); break;
    default: v = make_float4(1.f, 1.f, 1.f, 1.f); break;
  }
  return v;
}

// LayerNorm of one row held by TPR lanes, one float4 each (the arithmetic of gemm_nt's EPI_BIAS_RES_LN / rowchain's rc_ln_row)
template <int TPR>
__device__ __forceinline__ float lr_ln_row(float4 x, float4 gm, float4 bt, float inv_n, float eps, float4& h, float4& o) {
  const float mean = group_sum<TPR>((x.x + x.y) + (x.z + x.w)) * inv_n;
  x.x -= mean; x.y -= mean; x.z -= mean; x.w -= mean;
  const float q = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  const float rstd = 1.0f / sqrtf(group_sum<TPR>(q) * inv_n + eps);
  h.x = x.x * rstd; h.y = x.y * rstd; h.z = x.z * rstd; h.w = x.w * rstd;
  o.x = h.x * gm.x + bt.x; o.y = h.y * gm.y + bt.y; o.z = h.z * gm.z + bt.z; o.w = h.w * gm.w + bt.w;
  return rstd;
}
// LayerNorm backward of one row (ln_bwd_kernel's arithmetic): y = d loss / d LN output -> d loss / d LN input; dg / db = this thread's
// share of d gamma / d beta
template <int TPR>
__device__ __forceinline__ float4 lr_ln_bwd_row(float4 y, float4 h, float4 gm, float r, float inv_d, float4& dg, float4& db) {
  dg = make_float4(y.x * h.x, y.y * h.y, y.z * h.z, y.w * h.w);
  db = y;
  float4 gy = make_float4(y.x * gm.x, y.y * gm.y, y.z * gm.z, y.w * gm.w);
  const float s1 = (gy.x + gy.y) + (gy.z + gy.w);
  const float s2 = (gy.x * h.x + gy.y * h.y) + (gy.z * h.z + gy.w * h.w);
  const float m1 = group_sum<TPR>(s1) * inv_d, m2 = group_sum<TPR>(s2) * inv_d;
  return make_float4(r * (gy.x - m1 - h.x * m2), r * (gy.y - m1 - h.y * m2), r * (gy.z - m1 - h.z * m2), r * (gy.w - m1 - h.w * m2));
}

template <int D, int HD, int RG>
struct LrGeom {
  static constexpr int R = 4 * RG;              // sequences per workgroup
  static constexpr int H = D / HD;              // heads
  static constexpr int VWD = D / 64;            // columns per lane of an N = D product (one 64-lane tile spans D)
  static constexpr int XS = D + 4;              // row stride of the [R][D] LDS tiles (rows 4 floats apart in bank space: the four A rows of a block do not collide)
  static constexpr int TPR = D / 4;             // lanes per row in the row-wise epilogues
  static constexpr int KND = D / LR_NW;         // k-rows per wave of a K = D, N = D product
  static constexpr int SD = KND / LrH<D>::Q;    // ... = load instructions per wave of such a product
  static constexpr int SI = LR_K2MAX / LrH<D>::Q;   // ... of a K = inner, N = D product (inner <= 512)
  static constexpr int NP = LrH<D>::NP;         // K-part partial tiles of an N = D product
  static constexpr int WPS = LR_NW / R;         // waves per sequence in the attention phases
  static constexpr int LPH = 64 / H;            // lanes per head in a wave
  static constexpr int KQ = WPS * LPH;          // key parts per sequence: lane (h, q) of wave s walks keys pad + s * LPH + q, + KQ, ...
  static constexpr int T = (LR_MAXL + KQ - 1) / KQ;   // keys per lane (upper bound)
  static_assert(D == 64 || D == 128, "lastrow: d in {64, 128}");
  static_assert(H >= 4 && H <= 64 && 64 % H == 0, "lastrow: 4..64 heads");
  static_assert(RG == 1 || RG == 2, "lastrow: 4 or 8 sequences per workgroup");
  static_assert(T * 2 * HD <= 160, "lastrow: the K / V rows of a lane's keys live in registers");
};

// L2 warm-up.  The weights of a layer were rewritten a moment ago (the K-major copies by the transpose launch, the weights themselves by
// the optimizer), so no XCD's L2 holds them, and the workgroups of a launch walk them in lockstep: every line is a miss to memory for all
// sixteen workgroups of an XCD at the same time, and a CU's 32 KB of L1 lines in flight per ~2 000-cycle miss is 16-19 B / clk -- a 256 KB
// fragment set took 13 000 cycles to arrive (phase stamps, profiles/r04_b_lastrow_stamps.txt) against 4 000 from a warm L2 (wstream_probe).
// Each workgroup therefore first TOUCHES its own 1 / nsl share of every matrix (nsl = workgroups per XCD, workgroup g sits on XCD g % 8):
// together they request every line once, all misses in flight at the same time, and the fragment loads that follow find the lines in L2.
__device__ __forceinline__ void lr_touch(const float* W, int rows, int rowlen, int ld, int slice, int nsl, int tid, float& sink) {
  const int total4 = rows * (rowlen / 4), per = (total4 + nsl - 1) / nsl, lo = slice * per, hi = min(total4, lo + per);
  const int r4 = rowlen / 4;
  for (int i = lo + tid; i < hi; i += LR_THREADS) {
    const float4 v = *(const float4*)(W + (long long)(i / r4) * ld + (i % r4) * 4);
    sink += v.x;
  }
}

// phase stamps (ur_debug_lr_trace: a device buffer of [2][1024][32] shader-clock readings, forward then backward; thread 0 of a workgroup)
#define LR_STAMP(i) do { if (a.trace && tid == 0) a.trace[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

__device__ __forceinline__ float lr_exp_or0(float m, float M) { return m == -INFINITY ? 0.f : __expf(m - M); }

// ===================================================================================================================== forward
template <int D, int HD, int RG>
__global__ __launch_bounds__(LR_THREADS) void lastrow_fwd_kernel(LastRowFwdArgs a) {
  using G = LrGeom<D, HD, RG>;
  constexpr int R = G::R, H = G::H, XS = G::XS, KND = G::KND, T = G::T, TPR = G::TPR, SD = G::SD, SI = G::SI, NP = G::NP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int IS = a.I + 4;
  float* xs = smem;                       // [R][XS] layer input rows (residual of the attention block)
  float* qs = xs + R * XS;                // [R][XS] queries
  float* cs = qs + R * XS;                // [R][XS] attention output
  float* as_ = cs + R * XS;               // [R][XS] a (input and residual of the feed-forward block)
  float* us = as_ + R * XS;               // [R][IS] act(h1)
  float* part = us + R * IS;              // K-part partial tiles: max(8 * R * D, KP1 * R * I)
  float* mrg = part + a.part_floats;      // attention: per (sequence, wave, head) softmax state
  float* vec = mrg + R * G::WPS * H * (HD + 2);   // bq | bo | g1 | b1ln | b2 | g2 | b2ln (D each) | b1 (I): requested at entry -- read where
                                          // they are used, behind a barrier, each is a round trip to memory of its own on the critical path
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: it goes into buffer-load scalar offsets)
  const int b0 = blockIdx.x * R;
  const float inv_n = 1.0f / (float)D;
  LR_STAMP(0);
  float sink = 0.f;
  {
    const int nsl = min(16, ((int)gridDim.x + 7) / 8), slice = ((int)blockIdx.x / 8) % nsl;
    lr_touch(a.wqT, D, D, a.ldq, slice, nsl, tid, sink);
    lr_touch(a.woT, D, D, D, slice, nsl, tid, sink);
    lr_touch(a.w1T, D, a.I, a.I, slice, nsl, tid, sink);
    lr_touch(a.w2T, a.I, D, D, slice, nsl, tid, sink);
  }
  // ---- requests in the order they are needed: the query projection's fragments (K-part = wave), the input rows (staged at once: the
  // memory counter is in-order, whatever is requested in front of them is waited for with them), then this lane's K / V rows, which
  // have the whole query projection to arrive
  fx4 wq[SD];
  lr_fetch_h<D, SD>(wq, a.wqT, a.ldq, wave * KND, KND, lane);
  // attention roles
  const int ar = wave / G::WPS, ah = lane % H, akq = (wave % G::WPS) * G::LPH + lane / H;
  const int ab = min(b0 + ar, a.B - 1);
  const long long arow0 = a.seq_base ? (long long)a.seq_base[ab] : (long long)ab * a.L;
  const int apad = a.seq_pad ? a.seq_pad[ab] : 0;
  const int* asq = a.seq + (long long)ab * a.L;
  float4 vreg = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nvec4 = 7 * TPR + a.I / 4;         // <= LR_THREADS (D <= 128, inner <= 512)
  if (tid < nvec4) {
    const int w = tid / TPR;
    const float* p = w == 0 ? a.bq : w == 1 ? a.bo : w == 2 ? a.g1 : w == 3 ? a.b1ln : w == 4 ? a.b2 : w == 5 ? a.g2 : w == 6 ? a.b2ln : a.b1 - 7 * D;
    vreg = *(const float4*)(p + (w < 7 ? (tid % TPR) * 4 : tid * 4));
  }
  for (int i = tid; i < R * TPR; i += LR_THREADS) {
    const int r = i / TPR, et = i % TPR, b = b0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < a.B) {
      const long long xr = a.xrow ? (long long)a.xrow[b] : (long long)b * a.xstride + a.xoff;
      v = *(const float4*)(a.x + xr * D + et * 4);
      if (a.x_out) *(float4*)(a.x_out + (long long)b * D + et * 4) = v;
    }
    *(float4*)(xs + r * XS + et * 4) = v;
  }
  if (tid < nvec4) *(float4*)(vec + tid * 4) = vreg;
  asm volatile("" ::"v"(sink));     // (the touched values are waited for here, with the input rows: the counter is in-order anyway)
  __builtin_amdgcn_sched_barrier(0);
  float kr[T][HD], vr[T][HD];
  int sid[T];
  {
    const float* base = a.qkv + arow0 * (3 * D) + ah * HD;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int j = min(apad + akq + t * G::KQ, a.L - 1);
      const float* kp = base + (long long)j * (3 * D) + D;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 k4 = *(const float4*)(kp + c), v4 = *(const float4*)(kp + D + c);
        kr[t][c] = k4.x; kr[t][c + 1] = k4.y; kr[t][c + 2] = k4.z; kr[t][c + 3] = k4.w;
        vr[t][c] = v4.x; vr[t][c + 1] = v4.y; vr[t][c + 2] = v4.z; vr[t][c + 3] = v4.w;
      }
      sid[t] = asq[j];
    }
  }
  const int any_id = lane < a.L ? asq[lane] : 0;
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  LR_STAMP(1);
  // ---- 1. q = x Wq^T + bq
  {
    fx4 acc[RG][4];
    lr_zero<4, RG>(acc);
    lr_mma_h<D, SD, RG>(acc, wq, xs, XS, wave * KND, KND, lane);
    lr_put_h<D, RG, false>(acc, part, wave, lane);
  }
  fx4 wo[SD];
  lr_fetch_h<D, SD>(wo, a.woT, D, wave * KND, KND, lane);
  __builtin_amdgcn_sched_barrier(0);
  LR_STAMP(2);
  __syncthreads();
  LR_STAMP(3);
  for (int i = tid; i < R * TPR; i += LR_THREADS) {
    const int r = i / TPR, et = i % TPR, b = b0 + r;
    float4 s = *(const float4*)(vec + et * 4);
#pragma unroll
    for (int kp = 0; kp < NP; ++kp) {
      const float4 p = *(const float4*)(part + (kp * R + r) * D + et * 4);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *(float4*)(qs + r * XS + et * 4) = s;
    if (b < a.B) *(float4*)(a.q_out + (long long)b * D + et * 4) = s;
  }
  __syncthreads();
  LR_STAMP(4);
  // ---- 2. one-query attention: lane (head h, key part q) walks its keys with a private running softmax; the parts of a head are
  // merged across the lanes of the wave (shuffles) and the waves of the sequence (LDS)
  {
    const bool literal = __ballot(any_id > 0) == 0ull;   // no valid key at all: the reference's softmax runs over every (masked) key
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 q4 = *(const float4*)(qs + ar * XS + ah * HD + c);
      q[c] = q4.x; q[c + 1] = q4.y; q[c + 2] = q4.z; q[c + 3] = q4.w;
    }
    const unsigned rk = mix32((unsigned)((ab * H + ah) * a.L + (a.L - 1)) ^ a.dkey);
    float sv[T];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) s = fmaf(q[c], kr[t][c], s);
      sv[t] = s;
    }
    if (literal) {   // (wave-uniform; the division of the reference's literal path stays out of the common one)
#pragma unroll
      for (int t = 0; t < T; ++t) sv[t] = apad + akq + t * G::KQ < a.L ? sv[t] / a.sqrt_hd + -10000.0f : -INFINITY;
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) sv[t] = (apad + akq + t * G::KQ < a.L && sid[t] > 0) ? sv[t] * a.scale : -INFINITY;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) m = fmaxf(m, sv[t]);
    float l = 0.f, o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int j = apad + akq + t * G::KQ;
      const float p = sv[t] == -INFINITY ? 0.f : __expf(sv[t] - m);
      l += p;
      const float pd = a.dthresh ? p * drop_mul(rk, (unsigned)j, a.dthresh, a.dscale) : p;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = fmaf(pd, vr[t][c], o[c]);
    }
#pragma unroll
    for (int msk = H; msk < 64; msk <<= 1) {   // the other key parts of this head in the wave
      const float m2 = __shfl_xor(m, msk, 64), l2 = __shfl_xor(l, msk, 64);
      const float M = fmaxf(m, m2), e1 = lr_exp_or0(m, M), e2 = lr_exp_or0(m2, M);
      l = l * e1 + l2 * e2;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = o[c] * e1 + __shfl_xor(o[c], msk, 64) * e2;
      m = M;
    }
    if (lane < H) {
      float* dst = mrg + ((ar * G::WPS + wave % G::WPS) * H + ah) * (HD + 2);
      dst[0] = m; dst[1] = l;
#pragma unroll
      for (int c = 0; c < HD; ++c) dst[2 + c] = o[c];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  LR_STAMP(5);
  // feed-forward 1 fragments: N = I in 256-column tiles (4 columns per lane), the waves left over split K
  const int nt1 = a.I / 256, kp1n = LR_NW / nt1, kn1 = D / kp1n;   // host: nt1 in {1, 2, 4, 8}
  const int t1 = wave % nt1, k1 = wave / nt1;
  fx4 w1[D / 4];                                                     // up to D / 4 k-rows per wave (nt1 = 2 at D = 128: 32 rows)
  lr_fetch_rt<4, D / 4>(w1, a.w1T, a.I, t1 * 256, k1 * kn1, kn1, lane);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  LR_STAMP(6);
  for (int i = tid; i < R * H; i += LR_THREADS) {   // merge the waves of a sequence; ctx and the log-sum-exp
    const int r = i / H, h = i % H, b = b0 + r;
    const float* src = mrg + (r * G::WPS * H + h) * (HD + 2);
    float m = src[0], l = src[1], o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = src[2 + c];
#pragma unroll
    for (int w = 1; w < G::WPS; ++w) {
      const float* s2 = src + w * H * (HD + 2);
      const float m2 = s2[0], M = fmaxf(m, m2), e1 = lr_exp_or0(m, M), e2 = lr_exp_or0(m2, M);
      l = l * e1 + s2[1] * e2;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = o[c] * e1 + s2[2 + c] * e2;
      m = M;
    }
    const float inv_l = 1.0f / l;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      const float v = o[c] * inv_l;
      cs[r * XS + h * HD + c] = v;
      if (b < a.B) a.ctx[(long long)b * D + h * HD + c] = v;
    }
    if (b < a.B) a.lse[(long long)b * H + h] = m + __logf(l);
  }
  __syncthreads();
  LR_STAMP(7);
  // ---- 3. a = LN(drop(ctx Wo^T + bo) + x)
  {
    fx4 acc[RG][4];
    lr_zero<4, RG>(acc);
    lr_mma_h<D, SD, RG>(acc, wo, cs, XS, wave * KND, KND, lane);
    lr_put_h<D, RG, false>(acc, part, wave, lane);
  }
  __syncthreads();
  LR_STAMP(8);
  for (int i = tid; i < R * TPR; i += LR_THREADS) {
    const int r = i / TPR, et = i % TPR, b = b0 + r;
    float4 x = *(const float4*)(vec + D + et * 4);
#pragma unroll
    for (int kp = 0; kp < NP; ++kp) {
      const float4 p = *(const float4*)(part + (kp * R + r) * D + et * 4);
      x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
    }
    if (a.drop_out.thresh) x = drop4(x, drop_rowkey(a.drop_out, min(b, a.B - 1)), (unsigned)(et * 4), a.drop_out);
    const float4 rs = *(const float4*)(xs + r * XS + et * 4);
    x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w;
    float4 h, o;
    const float rstd = lr_ln_row<TPR>(x, *(const float4*)(vec + 2 * D + et * 4), *(const float4*)(vec + 3 * D + et * 4), inv_n, a.eps, h, o);
    *(float4*)(as_ + r * XS + et * 4) = o;
    if (b < a.B) {
      *(float4*)(a.ahat + (long long)b * D + et * 4) = h;
      *(float4*)(a.a + (long long)b * D + et * 4) = o;
      if (et == 0) a.rstd1[b] = rstd;
    }
  }
  __syncthreads();
  LR_STAMP(9);
  // ---- 4. h1 = a W1^T + b1, u = act(h1)
  {
    fx4 acc[RG][4];
    lr_zero<4, RG>(acc);
    lr_mma_rt<4, D / 4, RG>(acc, w1, as_, XS, k1 * kn1, kn1, lane);
    lr_put<4, RG>(acc, part, a.I, k1 * R, t1 * 256, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  LR_STAMP(10);
  // feed-forward 2 fragments: N = D, K = I over the 8 waves
  const int kn2 = a.I / LR_NW;                                       // host: I % 32 == 0, I <= 8 * LR_K2MAX
  fx4 w2[SI];
  lr_fetch_h<D, SI>(w2, a.w2T, D, wave * kn2, kn2, lane);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  LR_STAMP(11);
  for (int i = tid; i < R * (a.I / 4); i += LR_THREADS) {
    const int r = i / (a.I / 4), e4 = i % (a.I / 4), b = b0 + r;
    float4 s = *(const float4*)(vec + 7 * D + e4 * 4);
    for (int kp = 0; kp < kp1n; ++kp) {
      const float4 p = *(const float4*)(part + (long long)(kp * R + r) * a.I + e4 * 4);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    if (b < a.B) *(float4*)(a.h1 + (long long)b * a.I + e4 * 4) = s;
    *(float4*)(us + r * IS + e4 * 4) = lr_act_fwd4(s, a.act);
  }
  __syncthreads();
  LR_STAMP(12);
  // ---- 5. y = LN(drop(u W2^T + b2) + a)
  {
    fx4 acc[RG][4];
    lr_zero<4, RG>(acc);
    lr_mma_h<D, SI, RG>(acc, w2, us, IS, wave * kn2, kn2, lane);
    lr_put_h<D, RG, false>(acc, part, wave, lane);
  }
  __syncthreads();
  LR_STAMP(13);
  for (int i = tid; i < R * TPR; i += LR_THREADS) {
    const int r = i / TPR, et = i % TPR, b = b0 + r;
    float4 x = *(const float4*)(vec + 4 * D + et * 4);
#pragma unroll
    for (int kp = 0; kp < NP; ++kp) {
      const float4 p = *(const float4*)(part + (kp * R + r) * D + et * 4);
      x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
    }
    if (a.drop_ffn.thresh) x = drop4(x, drop_rowkey(a.drop_ffn, min(b, a.B - 1)), (unsigned)(et * 4), a.drop_ffn);
    const float4 rs = *(const float4*)(as_ + r * XS + et * 4);
    x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w;
    float4 h, o;
    const float rstd = lr_ln_row<TPR>(x, *(const float4*)(vec + 5 * D + et * 4), *(const float4*)(vec + 6 * D + et * 4), inv_n, a.eps, h, o);
    if (b < a.B) {
      *(float4*)(a.yhat + (long long)b * D + et * 4) = h;
      *(float4*)(a.y + (long long)b * D + et * 4) = o;
      if (et == 0) a.rstd2[b] = rstd;
    }
  }
  LR_STAMP(14);
}

// ===================================================================================================================== backward
// Phases (every [R][.] tile in LDS; K-part partial tiles summed by the epilogue threads (r, et) = tid, the same threads in every
// epilogue, so what one epilogue keeps in registers -- the unmasked g_tf, g_ta -- the next one still has):
//   0  g_tf = LNbwd(g_y)                                     -> g_tf (g_tfd), d gamma2 / d beta2 partial sums
//   1  g_h1 = (g_tf W2) * act'(h1)                            -> g_h1
//   2  g_ta = LNbwd(g_h1 W1 + g_tf)                           -> g_ta (g_tad), d gamma1 / d beta1 partial sums
//   3  g_ctx = g_ta Wo
//   4  one-query attention backward: lane (head, key part)    -> dK, dV rows of g_qkv (the weight-gradient GEMM reads them), dq, and
//      the [len][2H] tile S = [ds | p] of every sequence
//   5  x_L extra = dq Wq + g_ta;  V[u] = q_h Wk_h (u = h), g_ctx_h Wv_h (u = H + h): the K-parts of two d x d products, unreduced
//   6  g_x rows of the sequence = S V (+ the extra on row L-1)
template <int D, int HD, int RG>
__global__ __launch_bounds__(LR_THREADS) void lastrow_bwd_kernel(LastRowBwdArgs a) {
  using G = LrGeom<D, HD, RG>;
  constexpr int R = G::R, H = G::H, XS = G::XS, VWD = G::VWD, KND = G::KND, T = G::T, TPR = G::TPR, SD = G::SD, SI = G::SI, NP = G::NP;
  constexpr int SS = 2 * H + 4;           // row stride of the S tiles: four consecutive rows on four different bank quads
  constexpr int UPW = 2 * H / LR_NW;      // (q, head) / (g_ctx, head) units per wave in phase 5
  static_assert(2 * H % LR_NW == 0, "lastrow: 2 * heads must be a multiple of the wave count");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int IS = a.I + 4;
  float* gtf = smem;                      // [R][XS] g_tf (dropout-masked: the feed-forward branch's operand)
  float* gta = gtf + R * XS;              // [R][XS] g_ta (masked likewise)
  float* gc = gta + R * XS;               // [R][XS] g_ctx
  float* qs = gc + R * XS;                // [R][XS] q
  float* cs = qs + R * XS;                // [R][XS] ctx
  float* dqs = cs + R * XS;               // [R][XS] dq
  float* xl = dqs + R * XS;               // [R][XS] dq Wq + g_ta
  float* dqp = xl + R * XS;               // [R][WPS][D] per-wave partial dq
  float* part2 = dqp + R * G::WPS * D;    // [8][R][D] K-part partials of phase 5
  float* St = part2 + LR_NW * R * D;      // [R][LR_MAXL][SS]
  float* scr = St + R * LR_MAXL * SS;     // phases 0-3: part (K-part partials) | gh [R][IS] | red;  phases 5-6: V [R][2H][D]
  float* part = scr;
  float* gh = scr + a.part_floats;
  float* red = gh + R * IS;               // [R][2][D] column-sum scratch
  float* V = scr;
  float* aht = scr + a.scr_floats;        // [R][XS] ahat rows | [D] g1 | [R] rstd1: operands of the phase-2 epilogue, requested at entry
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= a.n_main) {   // riders: the zero-fills the backward pass starts with (its gradient buffers), 16 floats per thread
    const long long blk = (long long)blockIdx.x - a.n_main;
    if (blk == 0 && tid == 0 && a.copy_src) *a.copy_dst = *a.copy_src;
    const long long z1 = (a.zero_n + 8191) / 8192;
    const long long i0 = ((blk < z1 ? blk : blk - z1) * LR_THREADS + tid) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long e = i0 + q * 4;
      if (blk < z1) {
        if (e < a.zero_n) *(float4*)(a.zero_ptr + e) = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (e < a.zero2_n) {
        if (a.zero2_pad) {   // only the padded positions of every sequence: the valid rows are written whole by their producer
          const long long row = e / a.zero2_d;
          if ((int)(row % a.zero2_L) >= a.zero2_pad[row / a.zero2_L]) continue;
        }
        *(float4*)(a.zero2_ptr + e) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
  const int b0 = blockIdx.x * R;
  const float inv_d = 1.0f / (float)D;
  float* gpart = a.part + (long long)blockIdx.x * 4 * D;
  LR_STAMP(0);
  float sink = 0.f;
  {
    const int nsl = min(16, (a.n_main + 7) / 8), slice = ((int)blockIdx.x / 8) % nsl;
    lr_touch(a.w2, D, a.I, a.I, slice, nsl, tid, sink);
    lr_touch(a.w1, a.I, D, D, slice, nsl, tid, sink);
    lr_touch(a.wo, D, D, D, slice, nsl, tid, sink);
    lr_touch(a.wqkv, 3 * D, D, D, slice, nsl, tid, sink);
  }
  // ---- fragments of phase 1: N = I in 256-column tiles, the waves left over split K = D
  const int nt1 = a.I / 256, kp1n = LR_NW / nt1, kn1 = D / kp1n;
  const int t1 = wave % nt1, k1 = wave / nt1;
  fx4 w2f[D / 4];
  lr_fetch_rt<4, D / 4>(w2f, a.w2, a.I, t1 * 256, k1 * kn1, kn1, lane);
  // epilogue role of this thread: row er, float4 ee of it (threads beyond R * TPR idle in the row-wise epilogues)
  const bool erow = tid < R * TPR;
  const int er = erow ? tid / TPR : 0, ee = tid % TPR, eb = b0 + er;
  const bool evalid = erow && eb < a.B;
  const int ebc = min(eb, a.B - 1);
  float4 gtf_keep = make_float4(0.f, 0.f, 0.f, 0.f), gta_keep = gtf_keep;
  // (operands of the LATER epilogues, requested now: asked for where they are used they are a memory round trip each, behind a barrier)
  float4 q_ahat = make_float4(0.f, 0.f, 0.f, 0.f), q_g1 = q_ahat;
  float q_rstd1 = 0.f;
  if (erow) {
    q_ahat = *(const float4*)(a.ahat + (long long)ebc * D + ee * 4);
    q_g1 = *(const float4*)(a.g1 + ee * 4);
    q_rstd1 = a.rstd1[ebc];
  }
  // ---- 0. feed-forward LayerNorm backward
  {
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    if (erow) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (evalid) {
        const float4 y = *(const float4*)(a.gy + (long long)eb * D + ee * 4);
        const float4 h = *(const float4*)(a.yhat + (long long)eb * D + ee * 4);
        o = lr_ln_bwd_row<TPR>(y, h, *(const float4*)(a.g2 + ee * 4), a.rstd2[eb], inv_d, dg, db);
        *(float4*)(a.g_tf + (long long)eb * D + ee * 4) = o;
      } else {   // (keeps the row's lanes together in the reductions)
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        o = lr_ln_bwd_row<TPR>(z, z, z, 0.f, inv_d, dg, db);
      }
      gtf_keep = o;
      if (a.drop_ffn.thresh && evalid) {
        o = drop4(o, drop_rowkey(a.drop_ffn, eb), (unsigned)(ee * 4), a.drop_ffn);
        *(float4*)(a.g_tfd + (long long)eb * D + ee * 4) = o;
      }
      *(float4*)(gtf + er * XS + ee * 4) = o;
      *(float4*)(red + (er * 2 + 0) * D + ee * 4) = dg;
      *(float4*)(red + (er * 2 + 1) * D + ee * 4) = db;
      *(float4*)(aht + er * XS + ee * 4) = q_ahat;      // (parked in LDS: read back by this same thread in phase 2)
      if (er == 0) *(float4*)(aht + R * XS + ee * 4) = q_g1;
      if (ee == 0) aht[R * XS + D + er] = q_rstd1;
    }
  }
  asm volatile("" ::"v"(sink));     // (lr_touch: waited for with the loads of phase 0)
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  LR_STAMP(1);
  for (int i = tid; i < 2 * D; i += LR_THREADS) {   // d gamma2 | d beta2 of this workgroup's rows, fixed order
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += red[(r * 2 + i / D) * D + i % D];
    gpart[i] = s;
  }
  // ---- 1. g_h1 = (g_tf W2) * act'(h1)
  {
    fx4 acc[RG][4];
    lr_zero<4, RG>(acc);
    lr_mma_rt<4, D / 4, RG>(acc, w2f, gtf, XS, k1 * kn1, kn1, lane);
    lr_put<4, RG>(acc, part, a.I, k1 * R, t1 * 256, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  LR_STAMP(2);
  const int kn2 = a.I / LR_NW;
  fx4 w1f[SI];
  lr_fetch_h<D, SI>(w1f, a.w1, D, wave * kn2, kn2, lane);
  static_assert(R * 512 / 4 <= LR_THREADS, "the g_h1 epilogue is one float4 per thread (inner <= 512)");
  const bool hrow = tid < R * (a.I / 4);
  float4 q_h1 = make_float4(0.f, 0.f, 0.f, 0.f);   // this thread's h1 values of the epilogue below, requested ahead of the barrier
  if (hrow) q_h1 = *(const float4*)(a.h1 + (long long)min(b0 + tid / (a.I / 4), a.B - 1) * a.I + (tid % (a.I / 4)) * 4);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  LR_STAMP(3);
  if (hrow) {
    const int r = tid / (a.I / 4), e4 = tid % (a.I / 4), b = b0 + r;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kp = 0; kp < kp1n; ++kp) {
      const float4 p = *(const float4*)(part + (long long)(kp * R + r) * a.I + e4 * 4);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    const float4 da = lr_act_bwd4(q_h1, a.act);
    s.x *= da.x; s.y *= da.y; s.z *= da.z; s.w *= da.w;
    if (b < a.B) *(float4*)(a.g_h1 + (long long)b * a.I + e4 * 4) = s;
    *(float4*)(gh + r * IS + e4 * 4) = s;
  }
  __syncthreads();
  LR_STAMP(4);
  // ---- 2. g_a = g_h1 W1 + g_tf;  g_ta = LNbwd(g_a)
  {
    fx4 acc[RG][4];
    lr_zero<4, RG>(acc);
    lr_mma_h<D, SI, RG>(acc, w1f, gh, IS, wave * kn2, kn2, lane);
    lr_put_h<D, RG, false>(acc, part, wave, lane);
  }
  __builtin_amdgcn_sched_barrier(0);   // (the requests below stay behind the MFMAs above: hoisted, their registers overlap the fragments')
  LR_STAMP(5);
  fx4 wof[SD];
  lr_fetch_h<D, SD>(wof, a.wo, D, wave * KND, KND, lane);
  // attention roles; this lane's K / V rows and the staging of q / ctx are requested here
  const int ar = wave / G::WPS, ah = lane % H, akq = (wave % G::WPS) * G::LPH + lane / H;
  const int ab = min(b0 + ar, a.B - 1);
  const long long arow0 = a.seq_base ? (long long)a.seq_base[ab] : (long long)ab * a.L;
  const int apad = a.seq_pad ? a.seq_pad[ab] : 0;
  const int* asq = a.seq + (long long)ab * a.L;
  float kr[T][HD], vr[T][HD];
  int sid[T];
  {
    const float* base = a.qkv + arow0 * (3 * D) + ah * HD;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int j = min(apad + akq + t * G::KQ, a.L - 1);
      const float* kp = base + (long long)j * (3 * D) + D;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 k4 = *(const float4*)(kp + c), v4 = *(const float4*)(kp + D + c);
        kr[t][c] = k4.x; kr[t][c + 1] = k4.y; kr[t][c + 2] = k4.z; kr[t][c + 3] = k4.w;
        vr[t][c] = v4.x; vr[t][c + 1] = v4.y; vr[t][c + 2] = v4.z; vr[t][c + 3] = v4.w;
      }
      sid[t] = asq[j];
    }
  }
  const int any_id = lane < a.L ? asq[lane] : 0;
  const float alse = a.lse[(long long)ab * H + ah];
  if (erow) {
    *(float4*)(qs + er * XS + ee * 4) = *(const float4*)(a.q + (long long)ebc * D + ee * 4);
    *(float4*)(cs + er * XS + ee * 4) = *(const float4*)(a.ctx + (long long)ebc * D + ee * 4);
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  LR_STAMP(6);
  {
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    if (erow) {
      float4 y = gtf_keep;
#pragma unroll
      for (int kp = 0; kp < NP; ++kp) {
        const float4 p = *(const float4*)(part + (kp * R + er) * D + ee * 4);
        y.x += p.x; y.y += p.y; y.z += p.z; y.w += p.w;
      }
      float4 o = lr_ln_bwd_row<TPR>(y, *(const float4*)(aht + er * XS + ee * 4), *(const float4*)(aht + R * XS + ee * 4), aht[R * XS + D + er], inv_d, dg, db);
      if (!evalid) { o = make_float4(0.f, 0.f, 0.f, 0.f); dg = o; db = o; }
      gta_keep = o;
      if (evalid) *(float4*)(a.g_ta + (long long)eb * D + ee * 4) = o;
      if (a.drop_out.thresh && evalid) {
        o = drop4(o, drop_rowkey(a.drop_out, eb), (unsigned)(ee * 4), a.drop_out);
        *(float4*)(a.g_tad + (long long)eb * D + ee * 4) = o;
      }
      *(float4*)(gta + er * XS + ee * 4) = o;
      *(float4*)(red + (er * 2 + 0) * D + ee * 4) = dg;
      *(float4*)(red + (er * 2 + 1) * D + ee * 4) = db;
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * D; i += LR_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += red[(r * 2 + i / D) * D + i % D];
    gpart[2 * D + i] = s;
  }
  LR_STAMP(7);
  // ---- 3. g_ctx = g_ta Wo
  {
    fx4 acc[RG][4];
    lr_zero<4, RG>(acc);
    lr_mma_h<D, SD, RG>(acc, wof, gta, XS, wave * KND, KND, lane);
    lr_put_h<D, RG, false>(acc, part, wave, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  LR_STAMP(8);
  if (erow) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kp = 0; kp < NP; ++kp) {
      const float4 p = *(const float4*)(part + (kp * R + er) * D + ee * 4);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *(float4*)(gc + er * XS + ee * 4) = s;
  }
  __syncthreads();
  LR_STAMP(9);
  // ---- 4. attention backward of the one query: p recomputed from q, K and the saved log-sum-exp
  {
    const bool literal = __ballot(any_id > 0) == 0ull;
    const float f = literal ? 1.0f / a.sqrt_hd : a.scale;
    float q[HD], g[HD];
    float Dh = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 q4 = *(const float4*)(qs + ar * XS + ah * HD + c), g4 = *(const float4*)(gc + ar * XS + ah * HD + c);
      const float4 o4 = *(const float4*)(cs + ar * XS + ah * HD + c);
      q[c] = q4.x; q[c + 1] = q4.y; q[c + 2] = q4.z; q[c + 3] = q4.w;
      g[c] = g4.x; g[c + 1] = g4.y; g[c + 2] = g4.z; g[c + 3] = g4.w;
      Dh = fmaf(g4.x, o4.x, Dh); Dh = fmaf(g4.y, o4.y, Dh); Dh = fmaf(g4.z, o4.z, Dh); Dh = fmaf(g4.w, o4.w, Dh);
    }
    const unsigned rk = mix32((unsigned)((ab * H + ah) * a.L + (a.L - 1)) ^ a.dkey);
    const bool bvalid = b0 + ar < a.B;
    float dq[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dq[c] = 0.f;
    float* srow = St + (ar * LR_MAXL + akq) * SS;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int j = apad + akq + t * G::KQ;
      if (j < a.L) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          s = fmaf(q[c], kr[t][c], s);
          dp = fmaf(g[c], vr[t][c], dp);
        }
        float pj;
        if (literal) pj = __expf(s / a.sqrt_hd + -10000.0f - alse);   // (wave-uniform branch)
        else pj = sid[t] > 0 ? __expf(s * a.scale - alse) : 0.f;
        const float mk = a.dthresh ? drop_mul(rk, (unsigned)j, a.dthresh, a.dscale) : 1.0f;
        const float ds = pj * (mk * dp - Dh) * f, pm = pj * mk;
        srow[t * G::KQ * SS + ah] = ds;
        srow[t * G::KQ * SS + H + ah] = pm;
        if (bvalid) {
          float* out = a.g_qkv + (arow0 + j) * (3 * D) + D + ah * HD;
#pragma unroll
          for (int c = 0; c < HD; c += 4) {
            *(float4*)(out + c) = make_float4(ds * q[c], ds * q[c + 1], ds * q[c + 2], ds * q[c + 3]);
            *(float4*)(out + D + c) = make_float4(pm * g[c], pm * g[c + 1], pm * g[c + 2], pm * g[c + 3]);
          }
        }
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, kr[t][c], dq[c]);
      }
    }
#pragma unroll
    for (int msk = H; msk < 64; msk <<= 1)
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] += __shfl_xor(dq[c], msk, 64);
    if (lane < H) {
      float* dst = dqp + (ar * G::WPS + wave % G::WPS) * D + ah * HD;
#pragma unroll
      for (int c = 0; c < HD; ++c) dst[c] = dq[c];
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // (behind the attention phase: its K / V rows are dead, the registers are free)
  LR_STAMP(10);
  // fragments of phase 5: Wq K-part of this wave; the Wk / Wv rows of this wave's (operand, head) units
  fx4 wqf[SD];
  lr_fetch_h<D, SD>(wqf, a.wqkv, D, wave * KND, KND, lane);
  typename LrVec<VWD>::T wuf[UPW][HD];
#pragma unroll
  for (int q = 0; q < UPW; ++q) {
    const int u = wave * UPW + q;                      // u < H: q_h Wk_h;  u >= H: g_ctx_h Wv_h
    const int hh = u % H;
    lr_fetch<VWD, HD>(wuf[q], a.wqkv + (long long)(u < H ? 1 : 2) * D * D, D, 0, hh * HD, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  LR_STAMP(11);
  if (erow) {
    float4 s = *(const float4*)(dqp + (er * G::WPS) * D + ee * 4);
#pragma unroll
    for (int w = 1; w < G::WPS; ++w) {
      const float4 p = *(const float4*)(dqp + (er * G::WPS + w) * D + ee * 4);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *(float4*)(dqs + er * XS + ee * 4) = s;
    if (evalid) *(float4*)(a.dq + (long long)eb * D + ee * 4) = s;
  }
  __syncthreads();
  LR_STAMP(12);
  // ---- 5. dq Wq (K-parts -> part2) and the per-head row vectors V[r][u][:] (one unit = one head's HD k-rows: already "a K-part")
  {
    fx4 acc[RG][4];
    lr_zero<4, RG>(acc);
    lr_mma_h<D, SD, RG>(acc, wqf, dqs, XS, wave * KND, KND, lane);
    lr_put_h<D, RG, true>(acc, part2, wave, lane);      // (sub-rows folded: LR_NW parts -- the LDS budget of this kernel is tight)
  }
#pragma unroll
  for (int q = 0; q < UPW; ++q) {
    const int u = wave * UPW + q, hh = u % H;
    fx4 acc[RG][VWD];
    lr_zero<VWD, RG>(acc);
    lr_mma<VWD, HD, RG>(acc, wuf[q], u < H ? qs : gc, XS, hh * HD, lane);
    lr_put<VWD, RG>(acc, V + u * D, 2 * H * D, 0, 0, lane);       // V[(4 g + v) * 2H * D + u * D + col]
  }
  __syncthreads();
  LR_STAMP(13);
  if (erow) {
    float4 s = gta_keep;
#pragma unroll
    for (int kp = 0; kp < LR_NW; ++kp) {
      const float4 p = *(const float4*)(part2 + (kp * R + er) * D + ee * 4);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *(float4*)(xl + er * XS + ee * 4) = s;
  }
  __syncthreads();
  LR_STAMP(14);
  // ---- 6. the sequence's input-gradient rows: [len][2H] x [2H][D], four rows per MFMA; the waves of a sequence alternate row groups
  {
    const bool bvalid = b0 + ar < a.B;
    const int len = a.L - apad;
    typename LrVec<VWD>::T vf[2 * H];
#pragma unroll
    for (int u = 0; u < 2 * H; ++u) vf[u] = *(const typename LrVec<VWD>::T*)(V + (ar * 2 * H + u) * D + VWD * lane);
    typename LrVec<VWD>::T xlast = *(const typename LrVec<VWD>::T*)(xl + ar * XS + VWD * lane);
    constexpr int JB = 4;   // row groups in flight per wave: JB * VWD independent accumulator chains (a 4x4x1 MFMA result is ~60 cycles away)
    for (int jg0 = wave % G::WPS; 4 * jg0 < len; jg0 += JB * G::WPS) {
      fx4 acc[JB][VWD];
#pragma unroll
      for (int e = 0; e < JB; ++e)
#pragma unroll
        for (int c = 0; c < VWD; ++c) acc[e][c] = fx4{0.f, 0.f, 0.f, 0.f};
      const float* sp = St + (ar * LR_MAXL + 4 * jg0 + (lane & 3)) * SS;
#pragma unroll
      for (int u = 0; u < 2 * H; u += 4) {
        float4 s4[JB];
#pragma unroll
        for (int e = 0; e < JB; ++e) s4[e] = *(const float4*)(sp + min(4 * e * G::WPS, LR_MAXL - 4 - 4 * jg0) * SS + u);   // (clamped: rows beyond the tile are not stored below)
#pragma unroll
        for (int e = 0; e < JB; ++e)
#pragma unroll
          for (int c = 0; c < VWD; ++c) acc[e][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(s4[e].x, lr_comp<VWD>(vf[u], c), acc[e][c], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < JB; ++e)
#pragma unroll
          for (int c = 0; c < VWD; ++c) acc[e][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(s4[e].y, lr_comp<VWD>(vf[u + 1], c), acc[e][c], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < JB; ++e)
#pragma unroll
          for (int c = 0; c < VWD; ++c) acc[e][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(s4[e].z, lr_comp<VWD>(vf[u + 2], c), acc[e][c], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < JB; ++e)
#pragma unroll
          for (int c = 0; c < VWD; ++c) acc[e][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(s4[e].w, lr_comp<VWD>(vf[u + 3], c), acc[e][c], 0, 0, 0);
      }
      if (bvalid) {
#pragma unroll
        for (int e = 0; e < JB; ++e)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int jl = 4 * (jg0 + e * G::WPS) + v;
            if (jl < len) {
              float* out = a.g_x + (arow0 + apad + jl) * D + VWD * lane;
              const bool last = jl == len - 1;
              if constexpr (VWD == 2) {
                float2 o = make_float2(acc[e][0][v], acc[e][1][v]);
                if (last) { o.x += xlast[0]; o.y += xlast[1]; }
                *(float2*)out = o;
              } else {
                float o = acc[e][0][v];
                if (last) o += xlast;
                *out = o;
              }
            }
          }
      }
    }
  }
  LR_STAMP(15);
}

// ===================================================================================================================== launchers
bool lastrow_shape_ok(int B, int L, int d, int H, int inner) {
  if (!(d == 64 || d == 128) || H <= 0 || d % H) return false;
  const int hd = d / H;
  if (!(hd == 4 || hd == 8 || hd == 16)) return false;
  if (L < 8 || L > LR_MAXL) return false;                        // (L >= 8: the partial-sum slots of the layer hold B / 4 workgroups)
  if (!(inner == 256 || inner == 512)) return false;             // phases 1 / 4: 256-column tiles and at most d / 4 k-rows per wave; K = inner: <= 64 k-rows per wave
  (void)B;
  return true;
}
bool lastrow_supported(int B, int L, int d, int H, int inner) { return (chain_set_enabled(-1) & CHAIN_LASTROW) && lastrow_shape_ok(B, L, d, H, inner); }
int lastrow_rows_per_block() { return 4; }

template <typename KernelT>
static void lr_set_lds(KernelT k, size_t bytes) {
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
static int lr_part_floats(int d, int inner) {   // K-part partial tiles: N = d products (8 parts) and the N = inner product (8 / (inner / 256) parts)
  const int R = 4, a1 = LR_NW * (256 / d) * R * d, a2 = (LR_NW / (inner / 256)) * R * inner;   // (N = d products: LR_NW * Q parts, LrH)
  return a1 > a2 ? a1 : a2;
}

#define LR_DISPATCH(KERNEL, ARGS, GRID, LDS)                                                         \
  do {                                                                                               \
    const int hd_ = d / H;                                                                           \
    if (d == 128 && hd_ == 8) LR_GO(KERNEL, 128, 8, ARGS, GRID, LDS);                                \
    else if (d == 128 && hd_ == 16) LR_GO(KERNEL, 128, 16, ARGS, GRID, LDS);                         \
    else if (d == 128 && hd_ == 4) LR_GO(KERNEL, 128, 4, ARGS, GRID, LDS);                           \
    else if (d == 64 && hd_ == 4) LR_GO(KERNEL, 64, 4, ARGS, GRID, LDS);                             \
    else if (d == 64 && hd_ == 8) LR_GO(KERNEL, 64, 8, ARGS, GRID, LDS);                             \
    else LR_GO(KERNEL, 64, 16, ARGS, GRID, LDS);                                                     \
  } while (0)
#define LR_GO(KERNEL, D_, HD_, ARGS, GRID, LDS)                                                      \
  do {                                                                                               \
    static size_t attr_ = 0;                                                                         \
    if (attr_ < (LDS)) { lr_set_lds(KERNEL<D_, HD_, 1>, (LDS)); attr_ = (LDS); }                     \
    UR_LAUNCH_EV((KERNEL<D_, HD_, 1>), dim3(GRID), dim3(LR_THREADS), (LDS), st, ARGS);               \
  } while (0)

static unsigned long long* g_lr_trace = nullptr;   // [2][1024][32] (ur_debug_lr_trace)

int lastrow_fwd(const LastRowFwdArgs& a0, int d, int H, hipStream_t st) {
  if (a0.B <= 0) return UR_OK;
  if (!lastrow_shape_ok(a0.B, a0.L, d, H, a0.I)) return fail(UR_ERR_UNSUPPORTED, "lastrow_fwd: B=%d L=%d d=%d H=%d inner=%d", a0.B, a0.L, d, H, a0.I);
  LastRowFwdArgs a = a0;
  const int R = 4, hd = d / H;
  a.part_floats = lr_part_floats(d, a.I);
  a.trace = (g_lr_trace && cdiv(a.B, R) <= 1024) ? g_lr_trace : nullptr;
  const size_t lds = sizeof(float) * ((size_t)4 * R * (d + 4) + (size_t)R * (a.I + 4) + a.part_floats + (size_t)R * (LR_NW / R) * H * (hd + 2) + 7 * d + a.I);
  ProfScope ps(PC_CHAIN_SMALL, st, 2.0 * a.B * d * (2.0 * d + 2.0 * a.I) + 4.0 * a.B * a.L * d, true);
  LR_DISPATCH(lastrow_fwd_kernel, a, cdiv(a.B, R), lds);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int lastrow_bwd(const LastRowBwdArgs& a0, int d, int H, hipStream_t st) {
  if (a0.B <= 0) return UR_OK;
  if (!lastrow_shape_ok(a0.B, a0.L, d, H, a0.I)) return fail(UR_ERR_UNSUPPORTED, "lastrow_bwd: B=%d L=%d d=%d H=%d inner=%d", a0.B, a0.L, d, H, a0.I);
  LastRowBwdArgs a = a0;
  const int R = 4;
  a.part_floats = lr_part_floats(d, a.I);
  a.trace = (g_lr_trace && cdiv(a.B, R) <= 1024) ? g_lr_trace + 1024 * 32 : nullptr;
  size_t scr = (size_t)a.part_floats + (size_t)R * (a.I + 4) + (size_t)R * 2 * d;
  const size_t vfl = (size_t)R * 2 * H * d;
  if (vfl > scr) scr = vfl;
  a.scr_floats = (int)scr;
  const size_t lds = sizeof(float) * ((size_t)7 * R * (d + 4) + (size_t)R * (LR_NW / R) * d + (size_t)LR_NW * R * d +
                                      (size_t)R * LR_MAXL * (2 * H + 4) + scr + (size_t)R * (d + 4) + d + R);
  ProfScope ps(PC_CHAIN_SMALL, st, 2.0 * a.B * d * (3.0 * d + 2.0 * a.I) + 8.0 * a.B * a.L * d + 4.0 * a.B * a.L * H * d, true);
  a.n_main = cdiv(a.B, R);
  const int riders = (a.zero_ptr ? cdiv(a.zero_n, 8192) : 0) + (a.zero2_ptr ? cdiv(a.zero2_n, 8192) : 0);
  if (!a.zero_ptr) a.zero_n = 0;
  if (!a.zero2_ptr) a.zero2_n = 0;
  LR_DISPATCH(lastrow_bwd_kernel, a, a.n_main + (riders > 0 || a.copy_src ? (riders > 0 ? riders : 1) : 0), lds);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

}  // namespace ur

// debug (not part of the ABI): on = 1 allocates the stamp buffer, every later lastrow launch of the process writes its phase stamps;
// on = 0 copies [2][1024][32] uint64 (forward, backward; shader clock) to host_out and frees it
extern "C" int ur_debug_lr_trace(int on, unsigned long long* host_out) {
  const size_t bytes = (size_t)2 * 1024 * 32 * sizeof(unsigned long long);
  if (on) {
    if (!ur::g_lr_trace) {
      UR_HIP(hipMalloc((void**)&ur::g_lr_trace, bytes));
      UR_HIP(hipMemset(ur::g_lr_trace, 0, bytes));
    }
    return UR_OK;
  }
  if (!ur::g_lr_trace) return UR_OK;
  UR_HIP(hipDeviceSynchronize());
  if (host_out) UR_HIP(hipMemcpy(host_out, ur::g_lr_trace, bytes, hipMemcpyDeviceToHost));
  UR_HIP(hipFree(ur::g_lr_trace));
  ur::g_lr_trace = nullptr;
  return UR_OK;
}
