// Full-item ranking for one_vs_all evaluation.
// Reference: Evaluator.evaluate_with_full_items (unirec/facility/evaluation/evaluator_abc.py:189-278) copies the table
// to numpy, computes user_emb @ item_emb.T on the CPU, masks each user's history in a Python loop and counts
// scores above the target with numba get_rank (unirec/facility/evaluation/onepos.py:20-31).  Here:
//   rank[b] = #{ n in [0,N) minus {target} : s_bn > s_bt }  -  #{ n in distinct(history_b U {0}) minus {target} : s_bn > s_bt }
// The first term is ONE fp32-MFMA GEMM [B,d] x [d,N] whose epilogue compares and counts (gemm_nt EPI_COUNT_GT): the
// [B,N] score matrix is never written.  The second term is a sparse gather-dot over each user's (sorted) history.
// Scores are compared in the un-normalised space  u.E_n + item_bias[n]  (user bias cancels, tau > 0 is monotone).
#include "common.h"
#include "kernels.h"

namespace ur {

constexpr int MAXV = 4;
static inline int pick_tpr(int d) {
  int d4 = d / 4, t = 4;
  while (t < d4 && t < 32) t <<= 1;
  return t;
}

template <int TPR>
__device__ __forceinline__ float row_dot(const float4* __restrict__ table, long long id, const float4 (&u)[MAXV], int d4, int t) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = t + k * TPR;
    if (c < d4) {
      const float4 e = table[id * d4 + c];
      s += (e.x * u[k].x + e.y * u[k].y) + (e.z * u[k].z + e.w * u[k].w);
    }
  }
  return group_sum<TPR>(s);
}

// thr[b] = u_b . E[target_b] + item_bias[target_b];  target_score[b] = (thr[b] + user_bias[user_b]) / tau
template <int TPR>
__global__ __launch_bounds__(256) void target_score_kernel(const float4* __restrict__ user_emb, const float4* __restrict__ table,
                                                           const long long* __restrict__ target, const long long* __restrict__ user_id,
                                                           const float* __restrict__ user_bias, const float* __restrict__ item_bias,
                                                           float tau, int B, int d4, float* __restrict__ thr,
                                                           float* __restrict__ target_score) {
  constexpr int groups = 256 / TPR;
  const int b = blockIdx.x * groups + threadIdx.x / TPR, t = threadIdx.x % TPR;
  if (b >= B) return;
  float4 u[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = t + k * TPR;
    u[k] = c < d4 ? user_emb[(long long)b * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const long long id = target[b];
  float s = row_dot<TPR>(table, id, u, d4, t);
  if (t == 0) {
    if (item_bias) s += item_bias[id];
    thr[b] = s;
    target_score[b] = (s + (user_bias ? user_bias[user_id[b]] : 0.f)) / tau;
  }
}

// One workgroup per row b.  counts[b] += #{n in [n_lo, n_hi) : pass}  -  #{n in distinct(hist_b U {0, target_b}) : pass}
// (the range part covers the GEMM's N tail; the history ranges are ascending, duplicates adjacent).
template <int TPR>
__global__ __launch_bounds__(256) void rank_adjust_kernel(const float4* __restrict__ user_emb, const float4* __restrict__ table,
                                                          const long long* __restrict__ target, const long long* __restrict__ user_id,
                                                          const long long* __restrict__ hist_ptr, const int* __restrict__ hist_sorted,
                                                          long long n_users, const float* __restrict__ item_bias,
                                                          const float* __restrict__ thr, long long n_lo, long long n_hi, int d4,
                                                          int* __restrict__ counts) {
  constexpr int groups = 256 / TPR;
  __shared__ int acc[2];
  const int b = blockIdx.x, g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  if (threadIdx.x < 2) acc[threadIdx.x] = 0;
  __syncthreads();
  float4 u[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = t + k * TPR;
    u[k] = c < d4 ? user_emb[(long long)b * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float th = thr[b];
  const long long tgt = target[b];
  int plus = 0, minus = 0;
  for (long long n = n_lo + g; n < n_hi; n += groups) {   // tail of the item range
    const float s = row_dot<TPR>(table, n, u, d4, t) + (item_bias ? item_bias[n] : 0.f);
    plus += (s > th) && n != tgt;
  }
  const long long uid = user_id ? user_id[b] : -1;
  const bool known = hist_ptr && uid >= 0 && uid < n_users;
  const long long hb = known ? hist_ptr[uid] : 0, he = known ? hist_ptr[uid + 1] : 0;
  bool has0 = false;
  for (long long q = hb + g; q < he; q += groups) {
    const long long n = hist_sorted[q];
    if (q > hb && hist_sorted[q - 1] == n) continue;      // distinct (group-uniform)
    const float s = row_dot<TPR>(table, n, u, d4, t) + (item_bias ? item_bias[n] : 0.f);
    minus += (s > th) && n != tgt;   // the target column is never counted by the first term
    has0 |= n == 0;
  }
  if (t == 0) {
    atomicAdd(&acc[0], plus);
    atomicAdd(&acc[1], minus);
  }
  // does the history already contain item 0 / the target?  (block-wide OR through LDS ints)
  __shared__ int flag0;
  if (threadIdx.x == 0) flag0 = 0;
  __syncthreads();
  if (t == 0 && has0) atomicOr(&flag0, 1);
  __syncthreads();
  if (g == 0) {   // item 0 is the padding row: the reference overwrites column 0 with the target's score
    int extra = 0;
    if (!flag0 && tgt != 0) {
      const float s = row_dot<TPR>(table, 0, u, d4, t) + (item_bias ? item_bias[0] : 0.f);
      extra += s > th;
    }
    if (t == 0) atomicAdd(&counts[b], acc[0] - acc[1] - extra);
  }
}

}  // namespace ur

using namespace ur;

extern "C" int ur_full_rank(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                            const int64_t* target, const int64_t* user_id, const int64_t* hist_ptr, const int32_t* hist_sorted,
                            int64_t n_users, const float* user_bias, const float* item_bias, float tau, int32_t* rank,
                            float* target_score, float* thr_ws, void* stream) {
  UR_REQUIRE(user_emb && item_table && target && rank && target_score && thr_ws, UR_ERR_ARG, "ur_full_rank: null pointer");
  UR_REQUIRE(B > 0 && d > 0 && d % 4 == 0 && d <= 512 && n_items > 0 && n_items < (1LL << 31), UR_ERR_ARG, "ur_full_rank: shape");
  UR_REQUIRE(tau > 0.f, UR_ERR_ARG, "ur_full_rank: tau must be > 0 (scores are compared un-normalised)");
  UR_REQUIRE(!hist_ptr || (hist_sorted && user_id), UR_ERR_ARG, "ur_full_rank: history needs user_id and hist_sorted");
  UR_REQUIRE(!user_bias || user_id, UR_ERR_ARG, "ur_full_rank: user_bias needs user_id");
  hipStream_t st = as_stream(stream);
  const int tpr = pick_tpr(d), groups = 256 / tpr, d4 = d / 4;
#define GO(T) hipLaunchKernelGGL((target_score_kernel<T>), dim3(cdiv(B, groups)), dim3(256), 0, st, (const float4*)user_emb,          \
                                 (const float4*)item_table, (const long long*)target, (const long long*)user_id, user_bias, item_bias,  \
                                 tau, B, d4, thr_ws, target_score)
  switch (tpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    default: GO(32); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  UR_HIP(hipMemsetAsync(rank, 0, sizeof(int32_t) * B, st));
  const int64_t n_main = (n_items / 128) * 128;
  if (n_main > 0) {
    GemmArgs g{};
    g.A = user_emb; g.lda = d; g.W = item_table; g.ldw = d; g.C = (float*)rank; g.ldc = 0; g.M = B; g.N = (int)n_main; g.K = d;
    g.bias = item_bias; g.aux = thr_ws; g.ldaux = 0; g.skip = (const long long*)target;
    int rc = gemm_nt(g, PRO_NONE, EPI_COUNT_GT, st);
    if (rc) return rc;
  }
#define GO(T) hipLaunchKernelGGL((rank_adjust_kernel<T>), dim3(B), dim3(256), 0, st, (const float4*)user_emb, (const float4*)item_table, \
                                 (const long long*)target, (const long long*)user_id, (const long long*)hist_ptr, hist_sorted,            \
                                 (long long)n_users, item_bias, thr_ws, (long long)n_main, (long long)n_items, d4, rank)
  switch (tpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    default: GO(32); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}
