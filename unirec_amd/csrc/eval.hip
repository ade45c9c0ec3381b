// Full-item ranking for one_vs_all evaluation.
// Reference: Evaluator.evaluate_with_full_items (unirec/facility/evaluation/evaluator_abc.py:189-278) copies the table
// to numpy, computes user_emb @ item_emb.T on the CPU, masks each user's history in a Python loop and counts
// scores above the target with numba get_rank (unirec/facility/evaluation/onepos.py:20-31).  Here:
//   rank[b] = #{ n in [0,N) minus {target} : s_bn > s_bt }  -  #{ n in distinct(history_b U {0}) minus {target} : s_bn > s_bt }
// The first term is a streaming fp32-MFMA contraction [B,d] x [d,N] that compares and counts in registers
// (rank_stream_kernel, d <= 128; wider d: gemm_nt with the EPI_COUNT_GT epilogue): the [B,N] score matrix is never
// written.  The second term is a sparse gather-dot over each user's (sorted) history.
// Scores are compared in the un-normalised space  u.E_n + item_bias[n]  (user bias cancels, tau > 0 is monotone).
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace ur {

constexpr long long UR_TOPK_CHUNK = 1LL << 20;   // items scored per pass of ur_full_topk ([B, chunk] fp32 scratch)
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
  const long long id = target[b];   // < 0: the target is not a row of this table (row-sharded catalogue): thr = 0
  float s = row_dot<TPR>(table, id < 0 ? 0 : id, u, d4, t);
  if (t == 0) {
    if (item_bias) s += item_bias[id < 0 ? 0 : id];
    thr[b] = id < 0 ? 0.f : s;
    if (target_score) target_score[b] = (s + (user_bias ? user_bias[user_id[b]] : 0.f)) / tau;
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
                                                          int* __restrict__ counts, long long excl) {
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
    if (excl >= 0 && excl != tgt) {   // a second row that is not an item (row-sharded table: the unused local row 1 of rank 0)
      const float s = row_dot<TPR>(table, excl, u, d4, t) + (item_bias ? item_bias[excl] : 0.f);
      extra += s > th;
    }
    if (t == 0) atomicAdd(&counts[b], acc[0] - acc[1] - extra);
  }
}


// ------------------------------------------------------------------------------------------------ streaming rank
// Workgroup = 512 threads = 8 waves (2 x 4), tile = 128 users x 256 items, K = d <= 128 in 32-wide chunks.
// The 128 x d user tile is loaded into LDS ONCE; the workgroup then streams its share of the item table through a
// double-buffered 256 x 32 LDS stage (one barrier per stage, global loads of stage s+1 in flight during the MFMAs of
// stage s).  The comparison "score > target score" happens on the MFMA accumulators in registers; per-row counts
// stay in registers for the whole stream and leave as ONE integer atomic per (row, wave) at the end.
// The target's own score is produced by the same MFMA sequence (a first "diagonal" tile whose item rows are the
// gathered targets), so it is bit-identical to the value the stream computes for the target's column and the strict
// ">" never counts the target itself.
// Grid: workgroup id b -> XCD b % 8.  All user tiles of one item split get the same XCD and adjacent ids, so the item
// rows they share are fetched from HBM into one L2 once.
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct RankArgs {
  const float* user_emb; const float* table; const float* item_bias; const float* user_bias;
  const long long* target; const long long* user_id;
  long long N; int B, d, splits; float tau;
  float* thr; float* target_score; int* counts;
  int mode;   // 0: thresholds from the diagonal tile, then count;  1: thresholds only (target < 0 -> 0);  2: count against the given thr;
              // 3: EMIT every (score, item) with score > thr[row] into the row's candidate list (top-k pruning), counts = list lengths
  float* cand_v; long long* cand_i; int cap;
};

template <int KC>
__global__ __launch_bounds__(512) void rank_stream_kernel(RankArgs a) {
  constexpr int BM = 128, BN = 256, LSA = 32 * KC + 4, LSW = 36;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [BM][LSA]
  float* Ws = As + BM * LSA;         // [2][BN][LSW]
  float* thr_s = Ws + 2 * BN * LSW;  // [BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int ntm = (a.B + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int mt = q % ntm, split = (q / ntm) * 8 + xcd;
  const int m0 = mt * BM;
  const long long ntn = (a.N + BN - 1) / BN, tps = (ntn + a.splits - 1) / a.splits;
  const long long t_begin = (long long)split * tps, t_end = a.mode == 1 ? t_begin : min(ntn, t_begin + tps);
  if (a.mode == 1 ? split != 0 : (t_begin >= t_end && (split != 0 || a.mode >= 2))) return;   // split 0 publishes thr / target_score

  // ---- user tile -> LDS (zero fill beyond B and beyond d)
  for (int idx = tid; idx < BM * 8 * KC; idx += 512) {
    const int row = idx / (8 * KC), k = (idx % (8 * KC)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + row < a.B && k < a.d) v = *(const float4*)(a.user_emb + (long long)(m0 + row) * a.d + k);
    *(float4*)(As + row * LSA + k) = v;
  }

  const int c4 = tid & 7, lrow = tid >> 3;   // stage copy: 8 threads x 16 B per 128-B row segment, 64 rows per pass
  const float* Wp[4];
  float4 rw[4];
  auto set_rows = [&](long long tile) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = lrow + 64 * p;
      long long n;
      if (tile < 0) n = max(a.target[min(m0 + (row & (BM - 1)), a.B - 1)], 0LL);   // diagonal tile: the targets' rows
      else n = min(tile * BN + row, a.N - 1);
      Wp[p] = a.table + n * a.d + c4 * 4;
    }
  };
  auto load_global = [&](int kc) {
    const int k = kc * 32 + c4 * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) rw[p] = k < a.d ? *(const float4*)(Wp[p] + kc * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *(float4*)(Ws + buf * BN * LSW + (lrow + 64 * p) * LSW + c4 * 4) = rw[p];
  };

  floatx16 acc[2][2];
  int cnt[2][16];
  float thr_r[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      cnt[i][r] = 0;
      thr_r[i][r] = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j][r] = 0.f;
    }

  const int frow = lane & 31, fk = 4 * (lane >> 5), lcol = lane & 31, lrow4 = 4 * (lane >> 5);
  long long tile = a.mode >= 2 ? t_begin : -1;
  if (a.mode >= 2) {   // thresholds given (computed by the shard that owns each target, with the same MFMA sequence)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        thr_r[i][r] = row < a.B ? a.thr[row] : INFINITY;
      }
  }
  set_rows(tile);
  load_global(0);
  store_lds(0);
  __syncthreads();
  int buf = 0;
  for (;;) {
    // per-column bias of this tile (consumed after the KC stages below)
    float bias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wc * 64 + j * 32 + lcol;
      if (tile < 0) {
        bias[j] = a.item_bias ? a.item_bias[max(a.target[min(m0 + (col & (BM - 1)), a.B - 1)], 0LL)] : 0.f;
      } else {
        const long long n = tile * BN + col;
        bias[j] = n < a.N ? (a.item_bias ? a.item_bias[n] : 0.f) : -INFINITY;   // columns beyond N never count
      }
    }
    const long long next_tile = tile < 0 ? t_begin : tile + 1;
    const bool has_next = next_tile < t_end;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      bool staged = true;
      if (kc + 1 < KC) load_global(kc + 1);
      else if (has_next) { set_rows(next_tile); load_global(0); }
      else staged = false;
      const float* Ab = As + (wr * 64 + frow) * LSA + kc * 32 + fk;
      const float* Wb = Ws + buf * BN * LSW + (wc * 64 + frow) * LSW + fk;
#pragma unroll
      for (int kk = 0; kk < 32; kk += 8) {
        float4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const float4*)(Ab + i * 32 * LSA + kk);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *(const float4*)(Wb + j * 32 * LSW + kk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
          }
      }
      if (staged) store_lds(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    // ---- epilogue on the accumulators: acc[i][j][r] is row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31
    if (tile < 0) {
      if (wc == wr) {   // waves holding the diagonal 64x64 blocks
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if ((r & 3) + 8 * (r >> 2) + lrow4 == lcol) thr_s[wr * 64 + i * 32 + lcol] = acc[i][i][r] + bias[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + lrow4;
          thr_r[i][r] = m0 + row < a.B ? thr_s[row] : INFINITY;
        }
      if (split == 0 && tid < BM && m0 + tid < a.B) {
        const int m = m0 + tid;
        a.thr[m] = (a.mode == 1 && a.target[m] < 0) ? 0.f : thr_s[tid];
        if (a.target_score) a.target_score[m] = (thr_s[tid] + (a.user_bias ? a.user_bias[a.user_id[m]] : 0.f)) / a.tau;
      }
    } else if (a.mode == 3) {
      // top-k pruning: the thresholds are (a lower bound of) each row's k-th best score, so almost nothing passes
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r] + bias[j];
            if (v > thr_r[i][r]) {
              const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + lrow4;
              const int slot = atomicAdd(a.counts + m, 1);
              if (slot < a.cap) {
                a.cand_v[(long long)m * a.cap + slot] = v;
                a.cand_i[(long long)m * a.cap + slot] = tile * BN + wc * 64 + j * 32 + lcol;
              }
            }
          }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) cnt[i][r] += (acc[i][j][r] + bias[j] > thr_r[i][r]) ? 1 : 0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (!has_next) break;
    tile = next_tile;
  }
  // ---- counts: sum over the 32 lanes (columns) that share each row, one atomic per (row, wave)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int v = cnt[i][r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + lrow4;
      if (lcol == 0 && m < a.B && v > 0) atomicAdd(a.counts + m, v);
    }
}

template <int KC>
static int launch_rank_stream(const RankArgs& a, hipStream_t st) {
  constexpr size_t lds = (size_t)(128 * (32 * KC + 4) + 2 * 256 * 36 + 128) * sizeof(float);
  static const hipError_t attr = hipFuncSetAttribute((const void*)rank_stream_kernel<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  const int ntm = cdiv(a.B, 128);
  hipLaunchKernelGGL((rank_stream_kernel<KC>), dim3(ntm * a.splits), dim3(512), lds, st, a);
  UR_LAUNCH_CHECK();
  return UR_OK;
}


// ------------------------------------------------------------------------------------------------ full-item top-k
// model.topk (unirec/model/base/recommender.py:149-197): scores of ALL items, the user's history set to -inf, torch.topk.
// Here in item chunks: gemm_nt writes the chunk's scores [B, C], mask_scores_kernel applies the history / padding-row
// mask, row_topk_kernel keeps the chunk's k best per row (radix select on the order-preserving integer image of the
// floats, then a bitonic sort of the k survivors); the per-chunk winners are merged by the same kernel at the end.

// S[b, n - c0] = -inf for n in history(b) U {0}, n in [c0, c0 + cn)
__global__ __launch_bounds__(256) void mask_scores_kernel(float* __restrict__ S, long long ld, int B, long long c0, long long cn,
                                                          const long long* __restrict__ user_id, const long long* __restrict__ hist_ptr,
                                                          const int* __restrict__ hist_sorted, long long n_users) {
  const int b = blockIdx.x;
  float* row = S + (long long)b * ld;
  if (threadIdx.x == 0 && c0 == 0) row[0] = -INFINITY;   // item 0 is the padding row
  if (!hist_ptr) return;
  const long long u = user_id[b];
  if (u < 0 || u >= n_users) return;
  const long long hb = hist_ptr[u], he = hist_ptr[u + 1];
  for (long long q = hb + threadIdx.x; q < he; q += 256) {
    const long long n = hist_sorted[q];
    if (n >= c0 && n < c0 + cn) row[n - c0] = -INFINITY;
  }
}

__device__ __forceinline__ unsigned f2key(float x) {   // ascending unsigned order == ascending float order
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One workgroup per row.  vals[n] (+ ids[n] or implicit id = id_base + index) -> the k largest, sorted by (value desc,
// id asc), written to out_vals / out_ids at out_off (missing entries: -inf / -1).  k <= 1024.
__global__ __launch_bounds__(256) void row_topk_kernel(const float* __restrict__ vals, long long ld, const long long* __restrict__ ids,
                                                       long long ld_ids, long long n, long long id_base, int k,
                                                       float* __restrict__ out_vals, long long* __restrict__ out_ids, long long out_ld,
                                                       long long out_off) {
  __shared__ int hist[256];
  __shared__ unsigned s_prefix, s_mask;
  __shared__ int s_need, s_cnt_gt, s_cnt_eq;
  __shared__ unsigned skey[1024];
  __shared__ long long sid[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* v = vals + (long long)b * ld;
  const long long* vid = ids ? ids + (long long)b * ld_ids : nullptr;
  const int kk = (int)min((long long)k, n);
  if (tid == 0) { s_prefix = 0u; s_mask = 0u; s_need = kk; }
  __syncthreads();
  // ---- radix select: after the 4 passes s_prefix is the key of the kk-th largest value
  for (int pass = 0; pass < 4 && kk > 0; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix, mask = s_mask;
    for (long long i = tid; i < n; i += 256) {
      const unsigned key = f2key(v[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int need = s_need, dgt = 255;
      for (; dgt > 0; --dgt) {
        if (hist[dgt] >= need) break;
        need -= hist[dgt];
      }
      s_need = need;   // rank of the wanted element inside bucket dgt
      s_prefix = prefix | ((unsigned)dgt << shift);
      s_mask = mask | (0xFFu << shift);
    }
    __syncthreads();
  }
  // ---- collect: everything above the threshold key, then as many equal keys as are still needed
  const unsigned thr = s_prefix;
  if (tid == 0) { s_cnt_gt = 0; s_cnt_eq = 0; }
  for (int i = tid; i < 1024; i += 256) { skey[i] = 0u; sid[i] = 0x7FFFFFFFFFFFFFFFLL; }   // padding sorts last
  __syncthreads();
  const int n_eq = s_need;   // how many values equal to the threshold belong to the top kk
  for (long long i = tid; i < n && kk > 0; i += 256) {
    const unsigned key = f2key(v[i]);
    if (key > thr) {
      const int pos = atomicAdd(&s_cnt_gt, 1);
      skey[pos] = key;
      sid[pos] = vid ? vid[i] : id_base + i;
    }
  }
  __syncthreads();
  const int n_gt = s_cnt_gt;   // == kk - n_eq
  for (long long i = tid; i < n && kk > 0; i += 256) {
    const unsigned key = f2key(v[i]);
    if (key == thr) {
      const int e = atomicAdd(&s_cnt_eq, 1);
      if (e < n_eq) {
        skey[n_gt + e] = key;
        sid[n_gt + e] = vid ? vid[i] : id_base + i;
      }
    }
  }
  __syncthreads();
  // ---- bitonic sort of the 1024 slots by (key desc, id asc)
  int np2 = 1;
  while (np2 < kk) np2 <<= 1;
  if (np2 < 2) np2 = 2;
  for (int size = 2; size <= np2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < np2 / 2; t += 256) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool desc_block = ((lo & size) == 0);   // first half of each `size` block sorted "best first"
        const unsigned ka = skey[lo], kb = skey[hi];
        const long long ia = sid[lo], ib = sid[hi];
        const bool a_first = ka > kb || (ka == kb && ia < ib);   // a is better than b
        if (a_first != desc_block) { skey[lo] = kb; skey[hi] = ka; sid[lo] = ib; sid[hi] = ia; }
      }
      __syncthreads();
    }
  for (int i = tid; i < k; i += 256) {
    float val = -INFINITY;
    long long id = -1;
    if (i < kk) {
      const unsigned key = skey[i];
      val = __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
      id = sid[i];
    }
    out_vals[(long long)b * out_ld + out_off + i] = val;
    out_ids[(long long)b * out_ld + out_off + i] = id;
  }
}

// scores of a few trailing items the GEMM cannot take (N % 4): S[b, col0 + t] = u_b . E[n0 + t] + bias
__global__ __launch_bounds__(256) void tail_scores_kernel(const float* __restrict__ user_emb, const float* __restrict__ table,
                                                          const float* __restrict__ item_bias, int B, int d, long long n0, int nt,
                                                          float* __restrict__ S, long long ld, long long col0) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * nt) return;
  const int b = idx / nt, t = idx % nt;
  float s = 0.f;
  for (int c = 0; c < d; ++c) s = fmaf(user_emb[(long long)b * d + c], table[(n0 + t) * d + c], s);
  S[(long long)b * ld + col0 + t] = s + (item_bias ? item_bias[n0 + t] : 0.f);
}

// final (score + user_bias) / tau on the k winners (monotone: applied after the selection)
__global__ void topk_finish_kernel(float* __restrict__ vals, long long* __restrict__ ids, const long long* __restrict__ user_id,
                                   const float* __restrict__ user_bias, float tau, int B, int k) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * k) return;
  const int b = i / k;
  if (vals[i] != -INFINITY) vals[i] = (vals[i] + (user_bias ? user_bias[user_id[b]] : 0.f)) / tau;
  else ids[i] = -1;   // fewer than k admissible items: masked entries are reported as (-inf, -1)
}

// pruned top-k, step 1: thr[b] = (k-th best score of the first chunk) lowered by a rounding margin -- the first chunk was scored
// by gemm_nt, the stream accumulates in another order
__global__ void topk_threshold_kernel(const float* __restrict__ kth, long long ld, int B, float* __restrict__ thr) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float v = kth[(long long)b * ld];
  thr[b] = v == -INFINITY ? -INFINITY : v - (1e-4f * fabsf(v) + 1e-6f);
}
// pruned top-k, step 3: candidates that are the padding row or in the user's history leave (score = -inf)
__global__ __launch_bounds__(256) void cand_filter_kernel(float* __restrict__ cand_v, const long long* __restrict__ cand_i, int cap,
                                                          const int* __restrict__ counts, const long long* __restrict__ user_id,
                                                          const long long* __restrict__ hist_ptr, const int* __restrict__ hist_sorted,
                                                          long long n_users) {
  const int b = blockIdx.x;
  const int n = min(counts[b], cap);
  const long long u = (hist_ptr && user_id) ? user_id[b] : -1;
  const bool known = hist_ptr && u >= 0 && u < n_users;
  const long long hb = known ? hist_ptr[u] : 0, he = known ? hist_ptr[u + 1] : 0;
  for (int q = threadIdx.x; q < n; q += 256) {
    const long long id = cand_i[(long long)b * cap + q];
    bool drop = id == 0;
    if (!drop && he > hb) {
      long long lo = hb, hi = he;
      while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (hist_sorted[mid] < id) lo = mid + 1; else hi = mid;
      }
      drop = lo < he && hist_sorted[lo] == id;
    }
    if (drop) cand_v[(long long)b * cap + q] = -INFINITY;
  }
}

}  // namespace ur

using namespace ur;

// mode 0: the whole catalogue (ur_full_rank);  1 / 2: the two phases of the row-sharded count (ur_full_rank_shard)
static int full_rank_impl(int mode, const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                          const int64_t* target, const int64_t* user_id, const int64_t* hist_ptr, const int32_t* hist_sorted,
                          int64_t n_users, const float* user_bias, const float* item_bias, float tau, int32_t* rank,
                          float* target_score, float* thr_ws, void* stream, int64_t excl_row = -1) {
  UR_REQUIRE(user_emb && item_table && target && thr_ws && (mode == 1 || rank) && (mode != 0 || target_score), UR_ERR_ARG,
             "ur_full_rank: null pointer");
  UR_REQUIRE(B > 0 && d > 0 && d % 4 == 0 && d <= 512 && n_items > 0 && n_items < (1LL << 31), UR_ERR_ARG, "ur_full_rank: shape");
  UR_REQUIRE(tau > 0.f, UR_ERR_ARG, "ur_full_rank: tau must be > 0 (scores are compared un-normalised)");
  UR_REQUIRE(!hist_ptr || (hist_sorted && user_id), UR_ERR_ARG, "ur_full_rank: history needs user_id and hist_sorted");
  UR_REQUIRE(!user_bias || user_id, UR_ERR_ARG, "ur_full_rank: user_bias needs user_id");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_MISC, st, 2.0 * B * (double)n_items * d);
  const int tpr = pick_tpr(d), groups = 256 / tpr, d4 = d / 4;
  if (mode != 1) UR_HIP(hipMemsetAsync(rank, 0, sizeof(int32_t) * B, st));
  int64_t n_tail = n_items;   // items [n_tail, n_items) are left to the adjust kernel
  if (d <= 128) {
    // streaming kernel: user blocks of <= 4096 rows (32 user tiles x 8 item splits = 256 workgroups = one per CU)
    for (int b0 = 0; b0 < B; b0 += 4096) {
      RankArgs a{};
      a.B = std::min(4096, B - b0);
      a.user_emb = user_emb + (size_t)b0 * d; a.table = item_table; a.item_bias = item_bias; a.user_bias = user_bias;
      a.target = (const long long*)target + b0; a.user_id = user_id ? (const long long*)user_id + b0 : nullptr;
      a.N = n_items; a.d = d; a.tau = tau; a.mode = mode;
      a.thr = thr_ws + b0; a.target_score = target_score ? target_score + b0 : nullptr; a.counts = rank ? rank + b0 : nullptr;
      const int ntm = cdiv(a.B, 128);
      a.splits = 8 * std::max(1, 32 / ntm);
      int rc = d <= 32 ? launch_rank_stream<1>(a, st) : d <= 64 ? launch_rank_stream<2>(a, st)
             : d <= 96 ? launch_rank_stream<3>(a, st) : launch_rank_stream<4>(a, st);
      if (rc) return rc;
    }
  } else {
    if (mode != 2) {
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
    }
    if (mode == 1) return UR_OK;
    // generic GEMM with the count epilogue, in item chunks that keep the grid below HIP's 2^32-thread limit
    n_tail = (n_items / 128) * 128;
    const int64_t tiles_m8 = 8 * (int64_t)cdiv(cdiv(B, 128), 8);
    const int64_t chunk = std::max<int64_t>(1, (1LL << 23) / tiles_m8) * 128;
    for (int64_t n0 = 0; n0 < n_tail; n0 += chunk) {
      GemmArgs g{};
      g.A = user_emb; g.lda = d; g.W = item_table + (size_t)n0 * d; g.ldw = d; g.C = (float*)rank; g.ldc = 0; g.M = B;
      g.N = (int)std::min<int64_t>(chunk, n_tail - n0); g.K = d;
      g.bias = item_bias ? item_bias + n0 : nullptr; g.aux = thr_ws; g.ldaux = 0; g.skip = (const long long*)target; g.skip_base = n0;
      int rc = gemm_nt(g, PRO_NONE, EPI_COUNT_GT, st);
      if (rc) return rc;
    }
  }
  if (mode == 1) return UR_OK;
#define GO(T) hipLaunchKernelGGL((rank_adjust_kernel<T>), dim3(B), dim3(256), 0, st, (const float4*)user_emb, (const float4*)item_table, \
                                 (const long long*)target, (const long long*)user_id, (const long long*)hist_ptr, hist_sorted,            \
                                 (long long)n_users, item_bias, thr_ws, (long long)n_tail, (long long)n_items, d4, rank, (long long)excl_row)
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

extern "C" int ur_full_rank(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                            const int64_t* target, const int64_t* user_id, const int64_t* hist_ptr, const int32_t* hist_sorted,
                            int64_t n_users, const float* user_bias, const float* item_bias, float tau, int32_t* rank,
                            float* target_score, float* thr_ws, void* stream) {
  UR_TRACE_SCOPE();
  return full_rank_impl(0, user_emb, item_table, n_items, B, d, target, user_id, hist_ptr, hist_sorted, n_users, user_bias, item_bias, tau,
                        rank, target_score, thr_ws, stream);
}

extern "C" int ur_full_rank_shard(int32_t phase, const float* user_emb, const float* shard_table, int64_t n_local, int32_t B, int32_t d,
                                  const int64_t* local_target, const int64_t* user_id, const int64_t* hist_ptr,
                                  const int32_t* hist_sorted_local, int64_t n_users, const float* item_bias_local, int64_t excl_row,
                                  float* thr, int32_t* rank_partial, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(phase == 1 || phase == 2, UR_ERR_ARG, "ur_full_rank_shard: phase=%d", phase);
  UR_REQUIRE(excl_row < n_local, UR_ERR_ARG, "ur_full_rank_shard: excl_row=%lld", (long long)excl_row);
  return full_rank_impl(phase, user_emb, shard_table, n_local, B, d, local_target, user_id, phase == 2 ? hist_ptr : nullptr,
                        phase == 2 ? hist_sorted_local : nullptr, n_users, nullptr, item_bias_local, 1.0f, rank_partial, nullptr, thr, stream, excl_row);
}

extern "C" int64_t ur_full_topk_workspace_bytes(int32_t B, int64_t n_items, int32_t k) {
  if (B <= 0 || n_items <= 0 || k <= 0) return UR_ERR_ARG;
  const long long chunk = std::min<long long>((n_items + 3) & ~3LL, UR_TOPK_CHUNK);
  const long long nchunks = (n_items + chunk - 1) / chunk;
  return (long long)B * chunk * 4 + (long long)B * nchunks * k * (4 + 8) + 1024 + (long long)B * 16 + 4096;
}

extern "C" int ur_full_topk(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d, int32_t k,
                            const int64_t* user_id, const int64_t* hist_ptr, const int32_t* hist_sorted, int64_t n_users,
                            const float* user_bias, const float* item_bias, float tau, float* topk_scores, int64_t* topk_ids, void* ws,
                            void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(user_emb && item_table && topk_scores && topk_ids && ws, UR_ERR_ARG, "ur_full_topk: null pointer");
  UR_REQUIRE(B > 0 && d > 0 && d % 4 == 0 && n_items > 0 && n_items < (1LL << 31), UR_ERR_ARG, "ur_full_topk: shape");
  UR_REQUIRE(k > 0 && k <= 1024, UR_ERR_UNSUPPORTED, "ur_full_topk: k=%d (1..1024)", k);
  UR_REQUIRE(tau > 0.f, UR_ERR_ARG, "ur_full_topk: tau must be > 0");
  UR_REQUIRE(!hist_ptr || (hist_sorted && user_id), UR_ERR_ARG, "ur_full_topk: history needs user_id and hist_sorted");
  UR_REQUIRE(!user_bias || user_id, UR_ERR_ARG, "ur_full_topk: user_bias needs user_id");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_MISC, st, 2.0 * B * (double)n_items * d);
  const long long chunk = std::min<long long>((n_items + 3) & ~3LL, UR_TOPK_CHUNK);
  const long long nchunks = (n_items + chunk - 1) / chunk;
  float* S = (float*)ws;                                               // [B, chunk]
  float* cand_v = S + (long long)B * chunk;                            // [B, nchunks * k]
  long long* cand_i = (long long*)(((uintptr_t)(cand_v + (long long)B * nchunks * k) + 15) & ~(uintptr_t)15);
  // ---- pruned path (catalogues of 256 K items and more): a first range goes through the score-matrix pipeline and yields a lower bound of every row's
  // k-th best score; then ALL items are streamed once by the ranking kernel in EMIT mode -- scores stay in the MFMA accumulators,
  // only the few that beat the bound are written -- and the k best of those candidates are the answer.  No [B, N] score ever
  // reaches HBM.  Falls back to the chunked path when a row's candidate list overflows (k * N / chunk too large).
  static const long long cap_env = ur_test_hook("topk_cap");   // test hook: force list overflows
  // size of the first range: large enough that few items beat its k-th best (expected survivors per row: k * N / first),
  // small enough that the score-matrix pipeline over it is a fraction of the streaming pass
  const long long first = std::min<long long>(chunk, std::max<long long>(65536, (n_items / 16) & ~3LL));
  const long long cap = cap_env > 0 ? cap_env
                                    : std::min<long long>(chunk / 4, std::max<long long>(4096, 8LL * k * ((n_items + first - 1) / first)));
  const bool prune = d <= 128 && n_items >= 262144 && B <= 4096 && (long long)k * 64 <= first;
  int* cnt = (int*)(((uintptr_t)(cand_i + (long long)B * nchunks * k) + 15) & ~(uintptr_t)15);   // [B] list lengths
  float* thr = (float*)(cnt + B);                                                                  // [B]
  // k best of items [c0, c0 + cn) -> slot `slot` of the per-chunk winners (or the final output when direct)
  auto topk_range = [&](long long c0, long long cn, long long slot, bool direct) -> int {
    const long long cn4 = cn & ~3LL;
    if (cn4 > 0) {
      GemmArgs g{};
      g.A = user_emb; g.lda = d; g.W = item_table + (size_t)c0 * d; g.ldw = d; g.C = S; g.ldc = (int)chunk; g.M = B; g.N = (int)cn4; g.K = d;
      g.bias = item_bias ? item_bias + c0 : nullptr;
      int rc = gemm_nt(g, PRO_NONE, item_bias ? EPI_BIAS : EPI_NONE, st);
      if (rc) return rc;
    }
    if (cn > cn4) {
      hipLaunchKernelGGL(tail_scores_kernel, dim3(cdiv((long long)B * (cn - cn4), 256)), dim3(256), 0, st, user_emb, item_table, item_bias, B, d,
                         c0 + cn4, (int)(cn - cn4), S, chunk, cn4);
      UR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(mask_scores_kernel, dim3(B), dim3(256), 0, st, S, chunk, B, c0, cn, (const long long*)user_id,
                       (const long long*)hist_ptr, hist_sorted, (long long)n_users);
    UR_LAUNCH_CHECK();
    float* ov = direct ? topk_scores : cand_v;
    long long* oi = direct ? (long long*)topk_ids : cand_i;
    hipLaunchKernelGGL(row_topk_kernel, dim3(B), dim3(256), 0, st, S, chunk, (const long long*)nullptr, 0LL, cn, c0, k, ov, oi,
                       direct ? (long long)k : nchunks * k, direct ? 0LL : slot * k);
    UR_LAUNCH_CHECK();
    return UR_OK;
  };
  auto topk_chunk = [&](long long ci) -> int { return topk_range(ci * chunk, std::min(chunk, n_items - ci * chunk), ci, nchunks == 1); };
  if (prune) {
    int rc = topk_range(0, first, 0, false);
    if (rc) return rc;
  } else {
    for (long long ci = 0; ci < nchunks; ++ci) {
      int rc = topk_chunk(ci);
      if (rc) return rc;
    }
  }
  if (prune) {
    // thresholds from the first chunk's k-th best (slot k-1 of its sorted winners; -inf when it had fewer than k admissible items)
    hipLaunchKernelGGL(topk_threshold_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, cand_v + (k - 1), nchunks * k, B, thr);
    UR_LAUNCH_CHECK();
    float* ev = S;                                                                       // [B, cap] -- the score chunk is dead now
    long long* ei = (long long*)(((uintptr_t)(S + (long long)B * cap) + 15) & ~(uintptr_t)15);   // [B, cap]
    UR_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * B, st));
    UR_HIP(hipMemsetD32Async((hipDeviceptr_t)ev, 0xFF800000u, (size_t)B * cap, st));   // -inf: unused slots never win
    RankArgs a{};
    a.B = B; a.user_emb = user_emb; a.table = item_table; a.item_bias = item_bias; a.N = n_items; a.d = d; a.tau = 1.f; a.mode = 3;
    a.thr = thr; a.counts = cnt; a.cand_v = ev; a.cand_i = ei; a.cap = (int)cap;
    const int ntm = cdiv(B, 128);
    a.splits = 8 * std::max(1, 32 / ntm);
    int rc = d <= 32 ? launch_rank_stream<1>(a, st) : d <= 64 ? launch_rank_stream<2>(a, st)
           : d <= 96 ? launch_rank_stream<3>(a, st) : launch_rank_stream<4>(a, st);
    if (rc) return rc;
    std::vector<int> h_cnt(B);
    UR_HIP(hipMemcpyAsync(h_cnt.data(), cnt, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    UR_HIP(hipStreamSynchronize(st));
    bool overflow = false;
    for (int b = 0; b < B; ++b) overflow |= h_cnt[b] > cap;
    if (!overflow) {
      hipLaunchKernelGGL(cand_filter_kernel, dim3(B), dim3(256), 0, st, ev, ei, (int)cap, cnt, (const long long*)user_id,
                         (const long long*)hist_ptr, hist_sorted, (long long)n_users);
      UR_LAUNCH_CHECK();
      hipLaunchKernelGGL(row_topk_kernel, dim3(B), dim3(256), 0, st, ev, cap, ei, cap, cap, 0LL, k, topk_scores, (long long*)topk_ids,
                         (long long)k, 0LL);
      UR_LAUNCH_CHECK();
    } else {   // a candidate list overflowed: the chunked path, from the start
      for (long long ci = 0; ci < nchunks; ++ci) {
        int rc2 = topk_chunk(ci);
        if (rc2) return rc2;
      }
      if (nchunks > 1) {
        hipLaunchKernelGGL(row_topk_kernel, dim3(B), dim3(256), 0, st, cand_v, nchunks * k, cand_i, nchunks * k, nchunks * k, 0LL, k,
                           topk_scores, (long long*)topk_ids, (long long)k, 0LL);
        UR_LAUNCH_CHECK();
      }
    }
  } else if (nchunks > 1) {
    hipLaunchKernelGGL(row_topk_kernel, dim3(B), dim3(256), 0, st, cand_v, nchunks * k, cand_i, nchunks * k, nchunks * k, 0LL, k,
                       topk_scores, (long long*)topk_ids, (long long)k, 0LL);
    UR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(topk_finish_kernel, dim3(cdiv((long long)B * k, 256)), dim3(256), 0, st, topk_scores, (long long*)topk_ids, (const long long*)user_id,
                     user_bias, tau, B, k);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
