// fullsoftmax training loss: every item is a candidate.
// Reference: BaseRecommender.forward with loss_type == 'fullsoftmax' (unirec/model/base/recommender.py:46-55: label = item_id,
// in_item_id = arange(n_items)) and reco_abc.py:266-270:  loss = mean_b( logsumexp_n s(b,n) - s(b, target_b) ),
// s(b,n) = (u_b . E_n + user_bias[user_b] + item_bias[n]) / tau  (clamped to +-score_clip when set), n over ALL rows incl. 0.
//
// Built from the two fp32-MFMA GEMM kernels plus column-wise softmax kernels, in item chunks of UR_FS_CHUNK rows; the
// scores live TRANSPOSED, ST[n - c0, b], so that every contraction is one of the existing GEMM forms:
//   forward   ST = E_chunk U^T                        gemm_nt(A = E_chunk [C,d], W = U [B4,d])      -> running (max, sum) per b
//   backward  ST <- dS^T = (softmax - onehot) * scale  (scores recomputed per chunk)
//             dE_chunk = dS^T U                        gemm_nt(A = dS^T [C,B4], W = U^T [d,B4])      -> dense table gradient
//             dU      += dS E_chunk                    gemm_tn(P = dS^T [C,B4], Q = E_chunk [C,d])   (deterministic split over items)
// B4 = B rounded up to a multiple of 4 (zero users: their dS is forced to 0).  The table gradient is DENSE ([N,d], the
// reference's own cost: every row moves); row 0 (padding_idx) is zeroed.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace ur {

constexpr long long UR_FS_CHUNK = 1LL << 20;
constexpr int FS_SPLIT_ROWS = 2048;   // rows of ST per partial (max, sum)

__device__ __forceinline__ float fs_score(float dot, float ib, float ub, float inv_tau, float clip, bool* clipped) {
  float v = (dot + ib + ub) * inv_tau;
  *clipped = false;
  if (clip > 0.f && (v > clip || v < -clip)) {
    *clipped = true;
    v = fminf(fmaxf(v, -clip), clip);
  }
  return v;
}

// U [B,d] -> Upad [B4,d] (zero rows beyond B) and UT [d,B4]
__global__ void fs_pad_transpose_kernel(const float* __restrict__ U, int B, int B4, int d, float* __restrict__ Upad, float* __restrict__ UT) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B4 * d) return;
  const int b = (int)(i / d), c = (int)(i % d);
  const float v = b < B ? U[(long long)b * d + c] : 0.f;
  Upad[i] = v;
  UT[(long long)c * B4 + b] = v;
}

// per (row split, column b): running max / sum-exp over the rows of this split.  thread = column.
__global__ __launch_bounds__(256) void fs_lse_partial_kernel(const float* __restrict__ ST, int C, int B, int B4,
                                                             const float* __restrict__ item_bias, long long c0,
                                                             const long long* __restrict__ user_id, const float* __restrict__ user_bias,
                                                             float inv_tau, float clip, float* __restrict__ part_m,
                                                             float* __restrict__ part_l) {
  const int b = blockIdx.x * 256 + threadIdx.x, sp = blockIdx.y;
  if (b >= B) return;
  const float ub = user_bias ? user_bias[user_id[b]] : 0.f;
  const int r0 = sp * FS_SPLIT_ROWS, r1 = min(C, r0 + FS_SPLIT_ROWS);
  float m = -INFINITY, l = 0.f;
  for (int r = r0; r < r1; ++r) {
    bool cl;
    const float v = fs_score(ST[(long long)r * B4 + b], item_bias ? item_bias[c0 + r] : 0.f, ub, inv_tau, clip, &cl);
    if (v > m) {
      l = l * __expf(m - v) + 1.f;
      m = v;
    } else {
      l += __expf(v - m);
    }
  }
  part_m[(long long)sp * B + b] = m;
  part_l[(long long)sp * B + b] = l;
}

// fold this chunk's partials (fixed order) into the running (max, sum) of every column
__global__ void fs_lse_combine_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l, int n_split, int B,
                                      float* __restrict__ run_m, float* __restrict__ run_l, int first) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float m = first ? -INFINITY : run_m[b], l = first ? 0.f : run_l[b];
  for (int s = 0; s < n_split; ++s) {
    const float pm = part_m[(long long)s * B + b], pl = part_l[(long long)s * B + b];
    const float nm = fmaxf(m, pm);
    if (nm == -INFINITY) continue;
    l = l * __expf(m - nm) + pl * __expf(pm - nm);
    m = nm;
  }
  run_m[b] = m;
  run_l[b] = l;
}

// lse[b] = m + log(l);  loss = mean_b(lse[b] - s_target[b])   (single block, fixed order)
__global__ __launch_bounds__(256) void fs_loss_kernel(const float* __restrict__ run_m, const float* __restrict__ run_l,
                                                      const float* __restrict__ s_target, int B, float* __restrict__ lse,
                                                      float* __restrict__ loss_out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float v = run_m[b] + logf(run_l[b]);
    lse[b] = v;
    acc += v - s_target[b];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += red[i];
    loss_out[0] = t / (float)B;
    loss_out[1] = (float)B;
    loss_out[2] = t != t ? -1.0f : 1.0f;   // update guard (see ur_gather_dot_loss_fwd)
  }
}

// ST[r,b] <- d loss / d (u_b . E_{c0+r}) = (softmax - onehot) * scale (0 where the score was clipped; 0 for columns >= B);
// d_item_bias[c0 + r] = sum_b of the same (one wave per row, fixed lane order + xor-shuffle sum)
__global__ __launch_bounds__(256) void fs_grad_kernel(float* __restrict__ ST, int C, int B, int B4, const float* __restrict__ item_bias,
                                                      long long c0, const long long* __restrict__ user_id,
                                                      const float* __restrict__ user_bias, const float* __restrict__ lse,
                                                      const long long* __restrict__ target, float inv_tau, float clip, float scale,
                                                      const float* __restrict__ d_loss, float* __restrict__ d_item_bias) {
  const int lane = threadIdx.x & 63;
  if (d_loss) scale *= d_loss[0];
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= C) return;
  const float ib = item_bias ? item_bias[c0 + r] : 0.f;
  float rs = 0.f;
  for (int b = lane; b < B4; b += 64) {
    float g = 0.f;
    if (b < B) {
      bool cl;
      const float v = fs_score(ST[(long long)r * B4 + b], ib, user_bias ? user_bias[user_id[b]] : 0.f, inv_tau, clip, &cl);
      if (!cl) g = (__expf(v - lse[b]) - (target[b] == c0 + r ? 1.f : 0.f)) * scale;
    }
    ST[(long long)r * B4 + b] = g;
    rs += g;
  }
  rs = wave_sum(rs);
  if (lane == 0 && d_item_bias) d_item_bias[c0 + r] = rs;
}

// s_t[b] = s(b, row trow[b]) of the rows this rank scores (0 where trow[b] < 0: another rank's).  One wave per column.
__global__ __launch_bounds__(256) void fs_target_score_kernel(const float* __restrict__ U, const float* __restrict__ rows,
                                                              const long long* __restrict__ trow, int B, int d,
                                                              const float* __restrict__ item_bias, const long long* __restrict__ user_id,
                                                              const float* __restrict__ user_bias, float inv_tau, float clip,
                                                              float* __restrict__ s_t) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const long long t = trow[b];
  if (t < 0) {
    if (lane == 0) s_t[b] = 0.f;
    return;
  }
  float acc = 0.f;
  for (int c = lane; c < d; c += 64) acc = fmaf(U[(long long)b * d + c], rows[t * d + c], acc);
  acc = wave_sum(acc);
  bool cl;
  if (lane == 0) s_t[b] = fs_score(acc, item_bias ? item_bias[t] : 0.f, user_bias ? user_bias[user_id[b]] : 0.f, inv_tau, clip, &cl);
}

// The W ranks' partials [W][3][BA] = (running max, sum-exp, target score) of every column over that rank's rows, folded in RANK order
// (the same order on every rank: identical lse everywhere) -> lse[BA]; loss_out = [mean over THIS rank's columns col0 .. col0 + Bown of
// lse - s_target, Bown, update guard].  Single block.
__global__ __launch_bounds__(256) void fs_combine_shards_kernel(const float* __restrict__ parts, int W, int BA, int col0, int Bown,
                                                                float* __restrict__ lse, float* __restrict__ loss_out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < BA; b += 256) {
    float m = -INFINITY, l = 0.f, st = 0.f;
    for (int r = 0; r < W; ++r) {
      const float* p = parts + (long long)r * 3 * BA;
      const float pm = p[b], pl = p[BA + b];
      st += p[2 * BA + b];
      const float nm = fmaxf(m, pm);
      if (nm == -INFINITY) continue;
      l = l * __expf(m - nm) + pl * __expf(pm - nm);
      m = nm;
    }
    const float v = m + logf(l);
    lse[b] = v;
    if (b >= col0 && b < col0 + Bown) acc += v - st;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += red[i];
    loss_out[0] = t / (float)Bown;
    loss_out[1] = (float)Bown;
    loss_out[2] = t != t ? -1.0f : 1.0f;
  }
}

__global__ void fs_axpy_kernel(const float* __restrict__ x, long long n, float* __restrict__ y, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = accumulate ? y[i] + x[i] : x[i];
}

struct FsWs {
  float *ST, *Upad, *UT, *part_m, *part_l, *run_m, *run_l, *s_t, *du_tmp, *tn_ws;
  long long floats;
};
static FsWs fs_carve(int B, int d, long long N, float* base) {
  FsWs w;
  long long o = 0;
  auto take = [&](long long n) {
    float* p = base ? base + o : nullptr;
    o += (n + 63) & ~63LL;
    return p;
  };
  const int B4 = (B + 3) & ~3;
  const long long C = std::min<long long>(N, UR_FS_CHUNK);
  const long long nsp = (C + FS_SPLIT_ROWS - 1) / FS_SPLIT_ROWS;
  w.ST = take(C * B4); w.Upad = take((long long)B4 * d); w.UT = take((long long)B4 * d);
  w.part_m = take(nsp * B); w.part_l = take(nsp * B); w.run_m = take(B); w.run_l = take(B); w.s_t = take(B);
  w.du_tmp = take((long long)B4 * d);
  w.tn_ws = take(gemm_tn_ws_floats((int)C, B4, d));
  w.floats = o;
  return w;
}

}  // namespace ur

using namespace ur;

extern "C" int64_t ur_full_softmax_workspace_bytes(int32_t B, int32_t d, int64_t n_items) {
  if (B <= 0 || d <= 0 || n_items <= 0) return UR_ERR_ARG;
  return fs_carve(B, d, n_items, nullptr).floats * (int64_t)sizeof(float);
}

static int fs_check(const float* user_emb, const float* table, int64_t N, int B, int d, const int64_t* target, const int64_t* user_id,
                    const float* user_bias, float tau, const char* who) {
  UR_REQUIRE(user_emb && table && target, UR_ERR_ARG, "%s: null pointer", who);
  UR_REQUIRE(B > 0 && d > 0 && d % 4 == 0 && N > 0 && N < (1LL << 31), UR_ERR_ARG, "%s: shape", who);
  UR_REQUIRE(tau != 0.f, UR_ERR_ARG, "%s: tau", who);
  UR_REQUIRE(!user_bias || user_id, UR_ERR_ARG, "%s: user_bias needs user_id", who);
  return UR_OK;
}

// scores of one chunk, transposed: ST[C, B4] = E_chunk U^T
static int fs_scores(const FsWs& w, const float* table, long long c0, long long C, int B4, int d, hipStream_t st) {
  GemmArgs g{};
  g.A = table + (size_t)c0 * d; g.lda = d; g.W = w.Upad; g.ldw = d; g.C = w.ST; g.ldc = B4; g.M = (int)C; g.N = B4; g.K = d;
  return gemm_nt(g, PRO_NONE, EPI_NONE, st);
}

extern "C" int ur_full_softmax_fwd(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                                   const int64_t* target, const int64_t* user_id, const float* user_bias, const float* item_bias,
                                   float tau, float score_clip, const float* target_score, float* lse, float* loss_out, void* ws,
                                   void* stream) {
  UR_TRACE_SCOPE();
  int rc = fs_check(user_emb, item_table, n_items, B, d, target, user_id, user_bias, tau, "ur_full_softmax_fwd");
  if (rc) return rc;
  UR_REQUIRE(target_score && lse && loss_out && ws, UR_ERR_ARG, "ur_full_softmax_fwd: null pointer");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_LOSS, st, 2.0 * B * (double)n_items * d);
  FsWs w = fs_carve(B, d, n_items, (float*)ws);
  const int B4 = (B + 3) & ~3;
  hipLaunchKernelGGL(fs_pad_transpose_kernel, dim3(cdiv((long long)B4 * d, 256)), dim3(256), 0, st, user_emb, B, B4, d, w.Upad, w.UT);
  UR_LAUNCH_CHECK();
  const long long chunk = std::min<long long>(n_items, UR_FS_CHUNK);
  for (long long c0 = 0; c0 < n_items; c0 += chunk) {
    const long long C = std::min(chunk, n_items - c0);
    if ((rc = fs_scores(w, item_table, c0, C, B4, d, st))) return rc;
    const int nsp = cdiv(C, FS_SPLIT_ROWS);
    hipLaunchKernelGGL(fs_lse_partial_kernel, dim3(cdiv(B, 256), nsp), dim3(256), 0, st, w.ST, (int)C, B, B4, item_bias, c0,
                       (const long long*)user_id, user_bias, 1.0f / tau, score_clip, w.part_m, w.part_l);
    UR_LAUNCH_CHECK();
    hipLaunchKernelGGL(fs_lse_combine_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, w.part_m, w.part_l, nsp, B, w.run_m, w.run_l,
                       c0 == 0 ? 1 : 0);
    UR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(fs_loss_kernel, dim3(1), dim3(256), 0, st, w.run_m, w.run_l, target_score, B, lse, loss_out);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// Forward over a SHARD of the catalogue: part3[3][B] = (max, sum-exp, target score) of every column over the n_rows rows this rank
// scores (target_row[b] = row of column b's positive inside them, -1: another rank's).  ur_full_softmax_combine_shards folds the
// all-gathered partials of the W ranks.
extern "C" int ur_full_softmax_fwd_shard(const float* user_emb, const float* shard_rows, int64_t n_rows, int32_t B, int32_t d,
                                         const int64_t* target_row, const int64_t* user_id, const float* user_bias,
                                         const float* item_bias_rows, float tau, float score_clip, float* part3, void* ws, void* stream) {
  UR_TRACE_SCOPE();
  int rc = fs_check(user_emb, shard_rows, n_rows, B, d, target_row, user_id, user_bias, tau, "ur_full_softmax_fwd_shard");
  if (rc) return rc;
  UR_REQUIRE(part3 && ws, UR_ERR_ARG, "ur_full_softmax_fwd_shard: null pointer");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_LOSS, st, 2.0 * B * (double)n_rows * d);
  FsWs w = fs_carve(B, d, n_rows, (float*)ws);
  const int B4 = (B + 3) & ~3;
  hipLaunchKernelGGL(fs_pad_transpose_kernel, dim3(cdiv((long long)B4 * d, 256)), dim3(256), 0, st, user_emb, B, B4, d, w.Upad, w.UT);
  UR_LAUNCH_CHECK();
  const long long chunk = std::min<long long>(n_rows, UR_FS_CHUNK);
  for (long long c0 = 0; c0 < n_rows; c0 += chunk) {
    const long long C = std::min(chunk, n_rows - c0);
    if ((rc = fs_scores(w, shard_rows, c0, C, B4, d, st))) return rc;
    const int nsp = cdiv(C, FS_SPLIT_ROWS);
    hipLaunchKernelGGL(fs_lse_partial_kernel, dim3(cdiv(B, 256), nsp), dim3(256), 0, st, w.ST, (int)C, B, B4, item_bias_rows, c0,
                       (const long long*)user_id, user_bias, 1.0f / tau, score_clip, w.part_m, w.part_l);
    UR_LAUNCH_CHECK();
    hipLaunchKernelGGL(fs_lse_combine_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, w.part_m, w.part_l, nsp, B, part3, part3 + B,
                       c0 == 0 ? 1 : 0);
    UR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(fs_target_score_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, user_emb, shard_rows, (const long long*)target_row, B, d,
                     item_bias_rows, (const long long*)user_id, user_bias, 1.0f / tau, score_clip, part3 + 2 * (long long)B);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_full_softmax_combine_shards(const float* parts, int32_t world, int32_t B_all, int32_t col0, int32_t B_own, float* lse,
                                              float* loss_out, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(parts && lse && loss_out, UR_ERR_ARG, "ur_full_softmax_combine_shards: null pointer");
  UR_REQUIRE(world > 0 && B_all > 0 && B_own > 0 && col0 >= 0 && col0 + B_own <= B_all, UR_ERR_ARG, "ur_full_softmax_combine_shards: shape");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(fs_combine_shards_kernel, dim3(1), dim3(256), 0, st, parts, world, B_all, col0, B_own, lse, loss_out);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// The same over a SHARD of the catalogue (facility/distributed.py: fullsoftmax over a row-sharded table): `item_table` / `item_bias` /
// `d_item_table` / `d_item_bias` are the n_items rows this rank scores, `target[b]` the row of column b's positive INSIDE them (-1: another
// rank's), `lse` the log-sum-exp over the WHOLE catalogue (all ranks' partials combined), B every rank's columns.  zero_row0 = 0: row 0
// of the shard is an item like any other (only global row 0 is the padding row).
static int fs_bwd_impl(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                       const int64_t* target, const int64_t* user_id, const float* user_bias, const float* item_bias,
                       float tau, float score_clip, const float* lse, const float* d_loss, float* d_user_emb,
                       float* d_item_table, float* d_item_bias, void* ws, void* stream, int zero_row0) {
  int rc = fs_check(user_emb, item_table, n_items, B, d, target, user_id, user_bias, tau, "ur_full_softmax_bwd");
  if (rc) return rc;
  UR_REQUIRE(lse && d_user_emb && d_item_table && ws, UR_ERR_ARG, "ur_full_softmax_bwd: null pointer");
  UR_REQUIRE(!item_bias || d_item_bias, UR_ERR_ARG, "ur_full_softmax_bwd: d_item_bias is required when item_bias is given");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_LOSS, st, 6.0 * B * (double)n_items * d);
  FsWs w = fs_carve(B, d, n_items, (float*)ws);
  const int B4 = (B + 3) & ~3;
  const float scale = 1.0f / ((float)B * tau);   // times *d_loss (device scalar, nullable) inside the kernel
  hipLaunchKernelGGL(fs_pad_transpose_kernel, dim3(cdiv((long long)B4 * d, 256)), dim3(256), 0, st, user_emb, B, B4, d, w.Upad, w.UT);
  UR_LAUNCH_CHECK();
  const long long chunk = std::min<long long>(n_items, UR_FS_CHUNK);
  for (long long c0 = 0; c0 < n_items; c0 += chunk) {
    const long long C = std::min(chunk, n_items - c0);
    if ((rc = fs_scores(w, item_table, c0, C, B4, d, st))) return rc;
    hipLaunchKernelGGL(fs_grad_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, w.ST, (int)C, B, B4, item_bias, c0, (const long long*)user_id,
                       user_bias, lse, (const long long*)target, 1.0f / tau, score_clip, scale, d_loss, d_item_bias);
    UR_LAUNCH_CHECK();
    GemmArgs g{};   // dE_chunk [C,d] = dS^T [C,B4] U [B4,d]
    g.A = w.ST; g.lda = B4; g.W = w.UT; g.ldw = B4; g.C = d_item_table + (size_t)c0 * d; g.ldc = d; g.M = (int)C; g.N = d; g.K = B4;
    if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
    // dU [B4,d] (+)= dS [B4,C] E_chunk [C,d]
    if ((rc = gemm_tn(w.ST, B4, item_table + (size_t)c0 * d, d, (int)C, B4, d, 0, 0, w.du_tmp, d, nullptr, w.tn_ws, st))) return rc;
    hipLaunchKernelGGL(fs_axpy_kernel, dim3(cdiv((long long)B * d, 256)), dim3(256), 0, st, w.du_tmp, (long long)B * d, d_user_emb,
                       c0 == 0 ? 0 : 1);
    UR_LAUNCH_CHECK();
  }
  if (zero_row0) UR_HIP(hipMemsetAsync(d_item_table, 0, sizeof(float) * d, st));   // padding_idx = 0 (reco_abc.py:168)
  return UR_OK;
}

extern "C" int ur_full_softmax_bwd(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                                   const int64_t* target, const int64_t* user_id, const float* user_bias, const float* item_bias,
                                   float tau, float score_clip, const float* lse, const float* d_loss, float* d_user_emb,
                                   float* d_item_table, float* d_item_bias, void* ws, void* stream) {
  UR_TRACE_SCOPE();
  return fs_bwd_impl(user_emb, item_table, n_items, B, d, target, user_id, user_bias, item_bias, tau, score_clip, lse, d_loss, d_user_emb,
                     d_item_table, d_item_bias, ws, stream, 1);
}

extern "C" int ur_full_softmax_bwd_shard(const float* user_emb, const float* shard_rows, int64_t n_rows, int32_t B, int32_t d,
                                         const int64_t* target_row, const int64_t* user_id, const float* user_bias, const float* item_bias_rows,
                                         float tau, float score_clip, const float* lse, const float* d_loss, float* d_user_emb,
                                         float* d_shard_rows, float* d_item_bias_rows, int32_t zero_row0, void* ws, void* stream) {
  UR_TRACE_SCOPE();
  return fs_bwd_impl(user_emb, shard_rows, n_rows, B, d, target_row, user_id, user_bias, item_bias_rows, tau, score_clip, lse, d_loss,
                     d_user_emb, d_shard_rows, d_item_bias_rows, ws, stream, zero_row0 ? 1 : 0);
}
