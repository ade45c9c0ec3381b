// Row-sparse embedding gradient: sort the looked-up ids, segment-reduce the row gradients in a fixed
// order (deterministic, no float atomics), and update only the touched rows.
//
//   ur_rows_plan   : ids -> (uniq_idx, seg_start, sorted_pos, n_uniq)        stable LSD radix sort
//   ur_rows_reduce : explicit rows + implicit (coef * vec) rows -> uniq_grad
//   ur_sparse_adam_rows / ur_lazy_adam_catchup / ur_lazy_adam_flush          row-wise Adam
//
// The sort works at wavefront granularity: every wave owns a contiguous chunk of CH keys and walks it
// 64 keys at a time, so stability follows from program order; ranks inside a 64-key group come from
// __ballot match masks (no LDS atomics in the scatter), digit bases from a tiny single-block scan.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace ur {

constexpr int CH = 1024;     // keys per wave
constexpr int RADIX = 256;   // 8-bit digits

__device__ __forceinline__ unsigned long long lanemask_lt() {
  const unsigned lane = threadIdx.x & 63;
  return lane ? (~0ull >> (64 - lane)) : 0ull;
}
// mask of active lanes holding the same 8-bit digit as this lane
__device__ __forceinline__ unsigned long long match_digit(unsigned dgt, bool active) {
  unsigned long long m = __ballot(active);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned long long bal = __ballot((dgt >> b) & 1u);
    m &= ((dgt >> b) & 1u) ? bal : ~bal;
  }
  return m;
}

// ---- round 4: the multi-launch sort (n > SMALL_N: C3's 154 K ids per batch) as passes + 2 launches.  Rounds 1-3 ran three launches per
// 8-bit pass (per-wave histogram, a ONE-workgroup scan of the [digit][wave] table, scatter) + build + three for the heads: 13 launches,
// 0.34 ms at n = 153 728.  Now:
//   radix_first_kernel : keys / positions built, the per-wave histogram of pass 0 counted, the tables of the later passes zeroed;
//   radix_pass_kernel  : a workgroup derives the bases of ITS four waves from the whole (L2-resident, [wave][digit]) table itself --
//                        digit totals + the prefix over the earlier waves, one coalesced sweep of 256 threads, then a 256-entry scan in
//                        LDS: no scan launch -- scatters as before (wave-level, stable) and COUNTS the next pass's histogram where the
//                        keys land (integer atomics on [wave of the destination][next digit]: order-independent totals);
//   plan_merge_heads_kernel: run heads in one launch (as the small-batch path).
constexpr int RCH = 1024;    // keys per workgroup of the radix_first / radix_pass kernels: thread t of wave w owns keys w * 256 + it * 64 + lane, it < 4

__global__ __launch_bounds__(256) void radix_first_kernel(const int* __restrict__ ids_a, long long n_a, const long long* __restrict__ ids_b,
                                                          long long n_b, unsigned* __restrict__ keys, int* __restrict__ vals, int W,
                                                          long long n_local, int nchunks, int* __restrict__ hist, int n_later,
                                                          unsigned* __restrict__ status, int n_status, long long n_rows, IdGuard gd) {
  __shared__ int cnt[RADIX];
  const int chunk = blockIdx.x;
  const long long n = n_a + n_b, base = (long long)chunk * RCH;
  cnt[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RCH / 256; ++it) {
    const long long i = base + it * 256 + threadIdx.x;
    if (i < n) {
      const long long id = ur_guard_id((i < n_a) ? (long long)ids_a[i] : ids_b[i - n_a], n_rows, gd);
      // row-sharded table: row `id` lives on rank id % W at local row id / W + 1 (local row 0 = the padding row of EVERY shard); sorting by
      // (owner, local row) makes every owner's requests contiguous
      const unsigned key = (W > 1) ? (id ? (unsigned)((id % W) * n_local + id / W + 1) : 0u) : (unsigned)id;
      keys[i] = key;
      vals[i] = (int)i;
      atomicAdd(&cnt[key & 0xFF], 1);
    }
  }
  __syncthreads();
  hist[(long long)chunk * RADIX + threadIdx.x] = cnt[threadIdx.x];
  for (int p = 1; p <= n_later; ++p) hist[((long long)p * nchunks + chunk) * RADIX + threadIdx.x] = 0;
  if (chunk == 0)   // the heads kernel's look-back words (plan_merge_heads_kernel)
    for (int i = threadIdx.x; i < n_status; i += 256) status[i] = 0u;
}

__global__ __launch_bounds__(256) void radix_pass_kernel(const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in, long long n,
                                                         int shift, int nchunks, const int* __restrict__ hist, unsigned* __restrict__ keys_out,
                                                         int* __restrict__ vals_out, int* __restrict__ hist_next) {
  __shared__ int cntw[4][RADIX];   // per wave: keys of every digit seen so far (phase A), then the global position of the wave's first such key
  __shared__ int tot[RADIX];
  __shared__ __attribute__((aligned(16))) int tot4[4][RADIX], pre4[4][RADIX];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, chunk = blockIdx.x;
  const long long base = (long long)chunk * RCH + w * 256;
#pragma unroll
  for (int q = 0; q < 4; ++q) cntw[q][threadIdx.x] = 0;
  // ---- phase A (a wave's 256 keys, in order): digit, rank among the equal digits of the 64-key group, keys of that digit before the group
  unsigned key[4];
  int val[4], off[4];
  bool act[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {     // (all four loads first)
    const long long i = base + it * 64 + lane;
    act[it] = i < n;
    key[it] = act[it] ? keys_in[i] : 0u;
    val[it] = act[it] ? vals_in[i] : 0;
  }
  // ---- bases: digit totals over every chunk and the running sum in front of THIS chunk.  Wave w sweeps the chunks c = w, w + 4, ...
  // of the [chunk][digit] table, a lane four digits at a time (one 16-byte load per chunk, a wave reads the chunk's 1 KB row), eight
  // chunks in flight; the four waves' integer partial sums meet in LDS (19 dependent batches of 4-byte loads per thread before:
  // the sweep was most of a pass)
  {
    int4 run = make_int4(0, 0, 0, 0), prw = make_int4(0, 0, 0, 0);
    const int4* h4 = (const int4*)hist + lane;
    for (int c0 = w; c0 < nchunks; c0 += 32) {
      int4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = c0 + 4 * u < nchunks ? h4[(long long)(c0 + 4 * u) * (RADIX / 4)] : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        run.x += v[u].x; run.y += v[u].y; run.z += v[u].z; run.w += v[u].w;
        if (c0 + 4 * u < chunk) { prw.x += v[u].x; prw.y += v[u].y; prw.z += v[u].z; prw.w += v[u].w; }
      }
    }
    *(int4*)&tot4[w][4 * lane] = run;
    *(int4*)&pre4[w][4 * lane] = prw;
  }
  __syncthreads();
  const unsigned long long lt = lanemask_lt();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const unsigned dgt = (key[it] >> shift) & 0xFF;
    const unsigned long long m = match_digit(dgt, act[it]);
    const int rank = __popcll(m & lt);
    off[it] = act[it] ? cntw[w][dgt] + rank : 0;
    __builtin_amdgcn_wave_barrier();
    if (act[it] && rank == 0) cntw[w][dgt] += __popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  const int pre = pre4[0][threadIdx.x] + pre4[1][threadIdx.x] + pre4[2][threadIdx.x] + pre4[3][threadIdx.x];
  tot[threadIdx.x] = tot4[0][threadIdx.x] + tot4[1][threadIdx.x] + tot4[2][threadIdx.x] + tot4[3][threadIdx.x];
  __syncthreads();
  if (w == 0) {
    const int a0 = tot[4 * lane], a1 = tot[4 * lane + 1], a2 = tot[4 * lane + 2], a3 = tot[4 * lane + 3];
    const int sum = a0 + a1 + a2 + a3;
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    const int ex = inc - sum;
    tot[4 * lane] = ex; tot[4 * lane + 1] = ex + a0; tot[4 * lane + 2] = ex + a0 + a1; tot[4 * lane + 3] = ex + a0 + a1 + a2;
  }
  __syncthreads();
  {   // ---- phase B: thread d turns the four waves' counts of digit d into their starting positions
    int sum = tot[threadIdx.x] + pre;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = cntw[q][threadIdx.x];
      cntw[q][threadIdx.x] = sum;
      sum += t;
    }
  }
  __syncthreads();
  // ---- phase C: scatter (and count the NEXT pass's histogram where the keys land)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (act[it]) {
      const int pos = cntw[w][(key[it] >> shift) & 0xFF] + off[it];
      keys_out[pos] = key[it];
      vals_out[pos] = val[it];
      if (hist_next) atomicAdd(&hist_next[(long long)(pos / RCH) * RADIX + ((key[it] >> (shift + 8)) & 0xFF)], 1);
    }
  }
}

// ---- segment heads: count per wave -> scan -> write (uniq_idx, seg_start)
// small batches (n <= SMALL_N ids: every training batch of the C2 / C4 / C5 shapes) take the chunk-sort path further down
constexpr int SMALL_N = 32768;
constexpr int SMALL_N_MID = SMALL_N;          // (named apart because the chunk-sort kernels are defined further up)

// ------------------------------------------------------------------------------- segment reduce
constexpr int MAXV = 4;
constexpr int UR_CATCHUP_BLOCKS = 1024;
constexpr int UR_BACKGROUND_BLOCKS = 256;
constexpr int LONG_SEG = 64;
constexpr int FU_ROWS = 4;        // rows_reduce_kernel<.., FUSED>: unique ids a lane group has in flight at a time   // runs longer than this (hot items under Zipfian ids) are reduced by the whole block

// acc += gradient row of lookup position p (explicit row, or coef * vec for the scorer's implicit candidate rows)
template <int TPR>
__device__ __forceinline__ void add_pos(float4 (&acc)[MAXV], long long p, long long n_a, const float4* __restrict__ rows_a,
                                        const float* __restrict__ coef_b, const float4* __restrict__ vec_b, int G, int d4, int t) {
  if (p < n_a) {
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        const float4 r = rows_a[p * d4 + c];
        acc[k].x += r.x; acc[k].y += r.y; acc[k].z += r.z; acc[k].w += r.w;
      }
    }
  } else {
    const long long pb = p - n_a;
    const float w = coef_b[pb];
    const long long row = pb / G;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        const float4 r = vec_b[row * d4 + c];
        acc[k].x = fmaf(w, r.x, acc[k].x); acc[k].y = fmaf(w, r.y, acc[k].y);
        acc[k].z = fmaf(w, r.z, acc[k].z); acc[k].w = fmaf(w, r.w, acc[k].w);
      }
    }
  }
}

// The walk of one lane group over its share of a LONG run (positions first, first + step, ...): four positions per trip with
// every load of the trip issued before the first add -- position, coefficient and row loads are unconditional (clamped
// indices, pointer selects), because a load behind a branch is waited for on the spot.  Same adds in the same order as add_pos
// (acc + r == fma(1, r, acc) exactly), so results do not depend on which walk ran.  Hot items under Zipfian ids: ~1500
// positions of one id per batch; one dependent position -> row round trip per add made this kernel 204 us of a 1.0 ms step.
template <int TPR, int KV>
__device__ __forceinline__ void long_walk(float4 (&acc)[MAXV], int first, int end, int step, const int* __restrict__ sorted_pos,
                                          long long n_a, const float4* __restrict__ rows_a, const float* __restrict__ coef_b,
                                          const float4* __restrict__ vec_b, int G, int d4, int t) {
  constexpr int U = KV == 1 ? 4 : 1;   // positions per trip (KV float4 registers each; more costs the common short-run path its occupancy)
  const float* cb = coef_b ? coef_b : (const float*)rows_a;   // (never used when there are no b positions: every p < n_a)
  const float4* vb = vec_b ? vec_b : rows_a;
  int pn[U];                           // positions of the NEXT trip: loaded one trip ahead, under the row loads of this one
#pragma unroll
  for (int u = 0; u < U; ++u) pn[u] = sorted_pos[min(first + u * step, end - 1)];
  for (int q0 = first; q0 < end; q0 += U * step) {
    bool ok[U];
    int pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = q0 + u * step < end;
      pp[u] = pn[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) pn[u] = sorted_pos[min(q0 + (U + u) * step, end - 1)];
    float w[U];
    const float4* src[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool is_a = pp[u] < n_a;
      const int pb = is_a ? 0 : pp[u] - (int)n_a;
      const float cw = cb[pb];
      w[u] = is_a ? 1.0f : cw;
      src[u] = is_a ? rows_a + (long long)pp[u] * d4 : vb + (long long)(pb / G) * d4;
    }
    float4 r[U][KV];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < KV; ++k) r[u][k] = src[u][min(t + k * TPR, d4 - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ok[u]) {
#pragma unroll
        for (int k = 0; k < KV; ++k)
          if (t + k * TPR < d4) {
            acc[k].x = fmaf(w[u], r[u][k].x, acc[k].x); acc[k].y = fmaf(w[u], r[u][k].y, acc[k].y);
            acc[k].z = fmaf(w[u], r[u][k].z, acc[k].z); acc[k].w = fmaf(w[u], r[u][k].w, acc[k].w);
          }
      }
  }
}

// Work of the sharded step that RIDES in a row-reduce launch instead of paying for a launch of its own (5.5 us each on a 0.58 ms step):
//   step flags (owner side, first workgroup): out4 <- the flags every rank put into slot 0 of its block of the received gradient rows
//                                             (= exchange.hip shard_step_flags_kernel; rows_a is that buffer);
//   flag rows  (requester side, last workgroup): this rank's [NaN, overflow, loss, 1] into slot 0 of every block of `out`
//                                             (= shard_flag_rows_kernel; the sums themselves never land in a slot 0).
struct ReduceRiders {
  float* sf_out4 = nullptr;
  const float* fr_loss = nullptr;
  const int* fr_flags = nullptr;
  const int* fr_guard = nullptr;   // this device's id guard: raised = the flag row says "NaN" (skipped on every rank)
  int fr_on = 0, world = 0, cap = 0;
};

// ur_rows_reduce_update: the row update as the reduce kernel's epilogue -- the sum of a row's gradients goes straight into the optimizer
// rule for that row instead of out to uniq_grad and back in (two of the nine row-array passes of reduce + update, and one dependent
// launch).  Only where nothing needs the row gradients in between: no global-norm clipping, one rank.
struct FusedUpdate {
  int on = 0;
  AdamK a{};
  float4* table = nullptr; float4* mom = nullptr; float4* var = nullptr;
  int* last = nullptr;
  const float* scale_dev = nullptr;
  const int* guard_dev = nullptr;
  // the NEXT batch's lazy replay riding in the same launch (workgroups nb_main .. gridDim - 1): `cold` = its rows this step does not touch
  // (disjoint from every row the update writes), brought to "after this step"; `hot` = its rows that are in this step's plan too: current
  // through the update -- unless the step is skipped, then they are replayed as well
  const int* cold_idx = nullptr; const int* cold_n = nullptr;
  const int* hot_idx = nullptr; const int* hot_n = nullptr;
  long long list_max = 0;
  int nb_main = 0;     // workgroups of the reduction itself (0: all of them)
};
template <int TPR, int MODE>
__device__ __forceinline__ void sparse_adam_body(const AdamK& a, float4* __restrict__ table, float4* __restrict__ mom,
                                                 float4* __restrict__ var, int* __restrict__ last_step,
                                                 const int* __restrict__ uniq_idx, const int* __restrict__ n_uniq_dev, long long n_max,
                                                 const float4* __restrict__ grad, int d4, const float* __restrict__ scale_dev,
                                                 int bid, int nblk, const int* __restrict__ guard_dev = nullptr);
template <int TPR>
__device__ __forceinline__ void fused_row_update(const FusedUpdate& fu, long long row, const float4 (&grad)[MAXV], int t, int d4, float scale,
                                                 float bc1, float bc2s);
// (one float4 per lane, the row's state already in registers: requested before the gradient walk, so the two chains of dependent round
// trips -- plan entry -> positions -> gradient rows, and row id -> stamp, w, m, v -- travel together)
template <int TPR>
__device__ __forceinline__ void fused_row_update1(const FusedUpdate& fu, long long row, float4 gr, int last, float4 w, float4 m, float4 v,
                                                  int t, int c, bool cin, int d4, float scale, float bc1, float bc2s);

// One lane group per unique id; positions are summed in sorted (= lookup) order.  Runs longer than LONG_SEG are
// handled by all groups of the block together: group g takes positions s+g, s+g+groups, ..., the partial sums are
// combined through LDS in group order -- still a fixed summation order, so results are bit-reproducible.
template <int TPR, bool FUSED = false>
__global__ __launch_bounds__(256) void rows_reduce_kernel(const int* __restrict__ uniq_idx, const int* __restrict__ seg_start,
                                                          const int* __restrict__ sorted_pos, const int* __restrict__ n_uniq_dev,
                                                          long long n, const float4* __restrict__ rows_a, long long n_a,
                                                          const float* __restrict__ coef_b, const float4* __restrict__ vec_b, int G,
                                                          int d4, float4* __restrict__ out, int zero_tail,
                                                          const int* __restrict__ out_rows, ReduceRiders rd = ReduceRiders(),
                                                          FusedUpdate fu = FusedUpdate()) {
  float fscale = 1.f, fbc1 = 1.f, fbc2s = 1.f;
  if constexpr (FUSED) {
    if (rd.sf_out4) {
      // owner side of the sharded step: the gradient scale IS the step's flags, which sit in slot 0 of every source block of the received
      // gradient rows -- every workgroup sums them itself (world x 16 bytes, source-rank order: the same value everywhere), the first
      // one also publishes them (what the unfused launch's rider does)
      __shared__ float s_scale;
      if (threadIdx.x == 0) {
        float nan = 0.f, ovf = 0.f, loss = 0.f;
        for (int q = 0; q < rd.world; ++q) {
          const float4 r = rows_a[(long long)q * rd.cap * d4];
          nan += r.x; ovf += r.y; loss += r.z;
        }
        float sc = (nan > 0.f || ovf > 0.f) ? -1.f : 1.f / (float)rd.world;
        if (blockIdx.x == 0) {
          rd.sf_out4[0] = sc;
          rd.sf_out4[1] = nan > 0.f ? __builtin_nanf("") : loss / (float)rd.world;
          rd.sf_out4[2] = nan;
          rd.sf_out4[3] = ovf;
        }
        if (fu.guard_dev && *fu.guard_dev) sc = -1.f;
        s_scale = sc;
      }
      __syncthreads();
      fscale = s_scale;
    } else {
      fscale = ur_step_scale(fu.scale_dev, fu.guard_dev);
    }
    if (fu.nb_main > 0 && (int)blockIdx.x >= fu.nb_main) {
      // replay workgroups: the next batch's rows take their zero-gradient steps up to and including this one (MODE 1 walks to step - 1)
      AdamK a2 = fu.a;
      a2.step = fu.a.step + 1;
      const int rb = (int)blockIdx.x - fu.nb_main, rn = (int)gridDim.x - fu.nb_main;
      if (fu.cold_idx) sparse_adam_body<TPR, 1>(a2, fu.table, fu.mom, fu.var, fu.last, fu.cold_idx, fu.cold_n, fu.list_max, nullptr, d4, nullptr, rb, rn);
      if (fscale < 0.f && fu.hot_idx)     // a skipped step updates nothing: its rows of the next batch are replayed like the others
        sparse_adam_body<TPR, 1>(a2, fu.table, fu.mom, fu.var, fu.last, fu.hot_idx, fu.hot_n, fu.list_max, nullptr, d4, nullptr, rb, rn);
      return;
    }
    if (fscale < 0.f) return;     // update guard (NaN loss, id guard, overflow): the whole step is skipped, nothing is written
    fbc1 = 1.f - powf(fu.a.b1, (float)fu.a.step);
    fbc2s = sqrtf(1.f - powf(fu.a.b2, (float)fu.a.step));
  }
  const int nblk = (FUSED && fu.nb_main > 0) ? fu.nb_main : (int)gridDim.x;   // (workgroups of the reduction)
  if (!FUSED && rd.sf_out4 && blockIdx.x == 0 && threadIdx.x == 0) {   // (source-rank order: every rank sums the same values in the same order)
    float nan = 0.f, ovf = 0.f, loss = 0.f;
    for (int q = 0; q < rd.world; ++q) {
      const float4 r = rows_a[(long long)q * rd.cap * d4];
      nan += r.x; ovf += r.y; loss += r.z;
    }
    rd.sf_out4[0] = (nan > 0.f || ovf > 0.f) ? -1.f : 1.f / (float)rd.world;
    rd.sf_out4[1] = nan > 0.f ? __builtin_nanf("") : loss / (float)rd.world;
    rd.sf_out4[2] = nan;
    rd.sf_out4[3] = ovf;
  }
  if (rd.fr_on && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < rd.world) {
    const float loss = rd.fr_loss ? rd.fr_loss[0] : 0.f;
    const float nan = ((rd.fr_loss && (rd.fr_loss[2] < 0.f || loss != loss)) || (rd.fr_guard && *rd.fr_guard)) ? 1.f : 0.f;
    out[(long long)threadIdx.x * rd.cap * d4] = make_float4(nan, (rd.fr_flags && (rd.fr_flags[0] & 1)) ? 1.f : 0.f, nan != 0.f ? 0.f : loss, 1.f);
  }
  // (these two kernels run beside the bottom layer's weight-gradient launch: their few memory instructions go first -- 0.601 -> 0.596 ms/step)
  __builtin_amdgcn_s_setprio(3);
  constexpr int groups = 256 / TPR;
  __shared__ float4 part[groups][MAXV * TPR];
  __shared__ int long_list[256];
  __shared__ int long_cnt;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int n_uniq = *n_uniq_dev;
  auto uid = [&](long long i) -> long long { return i; };
  // (pass 2's first candidate test is issued here, so that its loads travel with pass 1's instead of after them)
  const long long gstride = nblk;
  const long long uc0 = blockIdx.x + gstride * threadIdx.x;
  bool cand0 = false;
  if (uc0 < n_uniq) {
    const long long u0 = uid(uc0);
    cand0 = uniq_idx[u0] != 0 && seg_start[u0 + 1] - seg_start[u0] > LONG_SEG;
  }
  // ---- pass 1, fused update with one float4 per lane (d <= 4 TPR: every benchmark shape): FU_ROWS unique ids per lane group at a time.
  // The kernel is a chain of dependent round trips per row -- plan entry -> position -> gradient row, and row id -> stamp, w, m, v, random
  // 512-byte reads over tables of tens of GB -- so its time is round trips x rounds: here every load of a stage is issued for all
  // FU_ROWS rows before the first use (unconditional loads at clamped indices, see sparse_adam_body).  Same adds in the same order
  // (first position as fma(w, r, 0), the rest by long_walk), same update arithmetic: bit-identical to the one-row loop below.
  bool pass1_done = false;
  if constexpr (FUSED) {
    if (d4 <= TPR) {
      constexpr int U = FU_ROWS;
      const float* cb = coef_b ? coef_b : (const float*)rows_a;
      const float4* vb = vec_b ? vec_b : rows_a;
      const int pc = min(t, d4 - 1);
      for (long long e0 = ((long long)blockIdx.x * groups + g) * U; e0 < n_uniq; e0 += (long long)nblk * groups * U) {
        long long row[U];
        int sg[U], eg[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
          const long long u = min(e0 + i, (long long)n_uniq - 1);
          row[i] = uniq_idx[u]; sg[i] = seg_start[u]; eg[i] = seg_start[u + 1];
        }
        int last[U], p0[U];
        float4 w[U], m[U], v[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
          last[i] = fu.last ? fu.last[row[i]] : fu.a.step - 1;
          w[i] = fu.table[row[i] * d4 + pc]; m[i] = fu.mom[row[i] * d4 + pc]; v[i] = fu.var[row[i] * d4 + pc];
          p0[i] = sorted_pos[sg[i]];      // (a unique id has at least one position)
        }
        float cw[U];
        const float4* src[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
          const bool is_a = p0[i] < n_a;
          const int pb = is_a ? 0 : p0[i] - (int)n_a;
          const float c = cb[pb];
          cw[i] = is_a ? 1.0f : c;
          src[i] = is_a ? rows_a + (long long)p0[i] * d4 : vb + (long long)(pb / G) * d4;
        }
        float4 r0[U];
#pragma unroll
        for (int i = 0; i < U; ++i) r0[i] = src[i][pc];
#pragma unroll
        for (int i = 0; i < U; ++i) {
          if (e0 + i >= n_uniq || row[i] == 0 || eg[i] - sg[i] > LONG_SEG) continue;   // group-uniform (long runs: pass 2)
          float4 acc[MAXV];
          acc[0] = make_float4(fmaf(cw[i], r0[i].x, 0.f), fmaf(cw[i], r0[i].y, 0.f), fmaf(cw[i], r0[i].z, 0.f), fmaf(cw[i], r0[i].w, 0.f));
          if (eg[i] - sg[i] > 1) long_walk<TPR, 1>(acc, sg[i] + 1, eg[i], 1, sorted_pos, n_a, rows_a, coef_b, vec_b, G, d4, t);
          fused_row_update1<TPR>(fu, row[i], acc[0], last[i], w[i], m[i], v[i], t, pc, t < d4, d4, fscale, fbc1, fbc2s);
        }
      }
      pass1_done = true;
    }
  }
  // ---- pass 1: one lane group per unique id, neighbouring ids in one workgroup (coalesced plan reads).  Long runs are left out.
  for (long long base = (long long)blockIdx.x * groups; base < n && !pass1_done; base += (long long)nblk * groups) {
    if (base >= n_uniq && !zero_tail) break;   // block-uniform
    const long long ent = base + g;
    if (ent >= n) continue;
    if (ent >= n_uniq) {
      if (zero_tail) {
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
          const int c = t + k * TPR;
          if (c < d4) out[ent * d4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      continue;
    }
    const long long u = uid(ent);
    float4 acc[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long frow = uniq_idx[u];
    const long long orow = out_rows ? out_rows[u] : ent;   // (out_rows: the row of unique id u in the caller's layout, e.g. its exchange slot)
    // slot 0 of every block is the flag row's: the padding id maps there (block 0), and so do keys that an OVERFLOWING pack cut (their
    // slot_of_uniq entry is stale, initially 0) -- a sum written there after the rider would lose the very flag that reports the overflow
    if (rd.fr_on && (frow == 0 || orow % rd.cap == 0)) continue;
    float4 pw, pm, pv;
    int plast = 0;
    const bool pre = FUSED && d4 <= TPR;
    const int pc = min(t, d4 - 1);
    if constexpr (FUSED) {
      if (pre && frow != 0) {     // (unconditional loads at a clamped column: a load behind a per-lane test is waited for on the spot)
        plast = fu.last ? fu.last[frow] : fu.a.step - 1;
        pw = fu.table[frow * d4 + pc]; pm = fu.mom[frow * d4 + pc]; pv = fu.var[frow * d4 + pc];
      }
    }
    if (frow != 0) {
      const int s = seg_start[u], e = seg_start[u + 1];
      if (e - s > LONG_SEG) continue;                      // pass 2
      if (e - s >= 8 && d4 <= TPR)   // a medium run: the pipelined walk, this lane group alone (step 1)
        long_walk<TPR, 1>(acc, s, e, 1, sorted_pos, n_a, rows_a, coef_b, vec_b, G, d4, t);
      else
        for (int q = s; q < e; ++q) add_pos<TPR>(acc, sorted_pos[q], n_a, rows_a, coef_b, vec_b, G, d4, t);
    }
    if constexpr (FUSED) {
      if (frow != 0) {   // (group-uniform)
        if (pre) fused_row_update1<TPR>(fu, frow, acc[0], plast, pw, pm, pv, t, pc, t < d4, d4, fscale, fbc1, fbc2s);
        else fused_row_update<TPR>(fu, frow, acc, t, d4, fscale, fbc1, fbc2s);
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) out[orow * d4 + c] = acc[k];
    }
  }
  // ---- pass 2: the long runs, each by a whole workgroup.  Candidates are dealt out INTERLEAVED (thread j of workgroup b looks at
  // id b + gridDim * j): hot items sit next to each other in id order (Zipfian catalogues are numbered by popularity), and a
  // workgroup walks its long runs one after the other -- with neighbouring ids in one workgroup the first one carried the eight
  // hottest items alone.  The order of the list does not matter: every run's sum has its own fixed order.
  for (long long j0 = 0; blockIdx.x + gstride * j0 < n_uniq; j0 += 256) {   // block-uniform
    if (threadIdx.x == 0) long_cnt = 0;
    __syncthreads();
    const long long uc = blockIdx.x + gstride * (j0 + threadIdx.x);
    bool cand = cand0;
    if (j0 != 0) {
      cand = false;
      if (uc < n_uniq) {
        const long long uu = uid(uc);
        cand = uniq_idx[uu] != 0 && seg_start[uu + 1] - seg_start[uu] > LONG_SEG;
      }
    }
    if (cand) long_list[atomicAdd(&long_cnt, 1)] = (int)uc;
    __syncthreads();
    const int cnt = long_cnt;
    for (int li = 0; li < cnt; ++li) {
      const long long el_ = long_list[li], ul = uid(el_);
      const int sl = seg_start[ul], el = seg_start[ul + 1];
      float4 acc[MAXV];
#pragma unroll
      for (int k = 0; k < MAXV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      switch ((d4 + TPR - 1) / TPR) {   // float4 chunks per lane (block-uniform)
        case 1: long_walk<TPR, 1>(acc, sl + g, el, groups, sorted_pos, n_a, rows_a, coef_b, vec_b, G, d4, t); break;
        case 2: long_walk<TPR, 2>(acc, sl + g, el, groups, sorted_pos, n_a, rows_a, coef_b, vec_b, G, d4, t); break;
        case 3: long_walk<TPR, 3>(acc, sl + g, el, groups, sorted_pos, n_a, rows_a, coef_b, vec_b, G, d4, t); break;
        default: long_walk<TPR, 4>(acc, sl + g, el, groups, sorted_pos, n_a, rows_a, coef_b, vec_b, G, d4, t); break;
      }
#pragma unroll
      for (int k = 0; k < MAXV; ++k) part[g][k * TPR + t] = acc[k];
      __syncthreads();
      if (g == 0) {
        float4 tot[MAXV];
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
          const int c = t + k * TPR;
          tot[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c < d4) {
            float4 r = part[0][k * TPR + t];
            for (int gg = 1; gg < groups; ++gg) {
              const float4 x = part[gg][k * TPR + t];
              r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w;
            }
            tot[k] = r;
            if constexpr (!FUSED) {
              const long long orow_l = out_rows ? (long long)out_rows[ul] : el_;
              if (!(rd.fr_on && orow_l % rd.cap == 0)) out[orow_l * d4 + c] = r;
            }
          }
        }
        if constexpr (FUSED) fused_row_update<TPR>(fu, (long long)uniq_idx[ul], tot, t, d4, fscale, fbc1, fbc2s);
      }
      __syncthreads();
    }
    __syncthreads();   // long_cnt / long_list are rewritten by the next round
  }
}

// ------------------------------------------------------------------------------- row-wise Adam
// zero-gradient optimizer steps t = from+1 .. to applied to one element (reference semantics: the dense torch optimizer
// moves every row every step; here the missed steps are replayed when the row is next needed).
//   Adam / AdamW: the momentum keeps moving the weight; replayed exactly for up to LAZY_EXACT_STEPS steps (beyond that the
//                 update b1^j * m / (...) is below fp32 resolution and only the moments decay, in closed form); with
//                 weight_decay != 0 every step is replayed.
//   SGD, Adagrad: a zero gradient is a no-op (weight_decay == 0); RMSprop: only square_avg decays (closed form).
// 192: the remaining updates sum to lr * m / sqrt(v) * sum_{j > 192} 0.9005^j = 1.6e-8 of the FIRST replayed step -- below the
// fp32 resolution of the weight it is added to (the first step itself is <= lr); the moments keep their exact decay.
constexpr int LAZY_EXACT_STEPS = 192;
// One float4 of (w, m, v).  The per-step scalars (bias corrections) are computed once per step for the 4 elements.
__device__ __forceinline__ void lazy_replay4(float4& w, float4& m, float4& v, int from, int to, const AdamK& a) {
  const int k = to - from;
  if (k <= 0) return;
  const bool dead = (m.x == 0.f && m.y == 0.f && m.z == 0.f && m.w == 0.f) && (v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f);
  if (a.wd == 0.f && dead) return;   // never touched (or fully decayed): zero-gradient steps are no-ops
  if (a.algo >= UR_OPT_SGD) {
    if (a.wd == 0.f) {
      if (a.algo == UR_OPT_RMSPROP) {
        const float f2 = powf(a.b2, (float)k);
        v.x *= f2; v.y *= f2; v.z *= f2; v.w *= f2;
      }
      return;
    }
    for (int j = 0; j < k; ++j) {   // weight decay acts on every row every step: replay (cost grows with the gap)
      opt_elem(w.x, m.x, v.x, 0.f, a, 1.f, 1.f); opt_elem(w.y, m.y, v.y, 0.f, a, 1.f, 1.f);
      opt_elem(w.z, m.z, v.z, 0.f, a, 1.f, 1.f); opt_elem(w.w, m.w, v.w, 0.f, a, 1.f, 1.f);
    }
    return;
  }
  float b1t = powf(a.b1, (float)from), b2t = powf(a.b2, (float)from);
  if (a.wd == 0.f) {
    // No weight decay: with a zero gradient the moments only decay, m_j = m_0 b1^j and v_j = v_0 b2^j, so the replayed weight is
    //   w_J = w_0 - m_0 * sum_j c_j / (sqrt(v_0) d_j + eps),   c_j = lr b1^j / (1 - b1^(from+j)),  d_j = sqrt(b2^j / (1 - b2^(from+j)))
    // -- per step two scalars shared by the four elements and ONE fma + reciprocal + fma per element (instead of re-deriving
    // m, v, sqrt(v) every step).  This loop is what a long run pays for the reference's dense-Adam semantics: at N = 100 M a row
    // comes back every ~3 600 steps, so every looked-up row replays LAZY_EXACT_STEPS steps.
    const int exact = min(k, LAZY_EXACT_STEPS);
    const float4 sv = make_float4(__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y), __builtin_amdgcn_sqrtf(v.z), __builtin_amdgcn_sqrtf(v.w));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float p1 = 1.f, p2 = 1.f;
    for (int j = 0; j < exact; ++j) {
      b1t *= a.b1; b2t *= a.b2; p1 *= a.b1; p2 *= a.b2;
      const float c = a.lr * p1 * __builtin_amdgcn_rcpf(1.f - b1t);
      const float dj = __builtin_amdgcn_sqrtf(p2 * __builtin_amdgcn_rcpf(1.f - b2t));
      acc.x = fmaf(c, __builtin_amdgcn_rcpf(fmaf(sv.x, dj, a.eps)), acc.x);
      acc.y = fmaf(c, __builtin_amdgcn_rcpf(fmaf(sv.y, dj, a.eps)), acc.y);
      acc.z = fmaf(c, __builtin_amdgcn_rcpf(fmaf(sv.z, dj, a.eps)), acc.z);
      acc.w = fmaf(c, __builtin_amdgcn_rcpf(fmaf(sv.w, dj, a.eps)), acc.w);
    }
    w.x = fmaf(-m.x, acc.x, w.x); w.y = fmaf(-m.y, acc.y, w.y); w.z = fmaf(-m.z, acc.z, w.z); w.w = fmaf(-m.w, acc.w, w.w);
    const float f1 = powf(a.b1, (float)k), f2 = powf(a.b2, (float)k);
    m.x *= f1; m.y *= f1; m.z *= f1; m.w *= f1;
    v.x *= f2; v.y *= f2; v.z *= f2; v.w *= f2;
    return;
  }
  const int exact = k;   // weight decay couples the weight back into the moments: every step is replayed
  const float c1m = 1.f - a.b1, c2m = 1.f - a.b2;
  const bool decoupled = a.algo == UR_OPT_ADAMW;
  const float shrink = 1.f - a.lr * a.wd;
  for (int j = 0; j < exact; ++j) {
    b1t *= a.b1;
    b2t *= a.b2;
    // hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32: 1 ulp) instead of the IEEE-exact division and sqrt
    // sequences: a replayed step moves a weight by ~lr * b1^j, so the difference is ~1e-7 of an already tiny step, far inside
    // the parity tolerance, and the replay of a frequently revisited table (C3: every row every ~13 steps) is VALU-bound
    const float inv_s2 = __builtin_amdgcn_rsqf(1.f - b2t), step = a.lr * __builtin_amdgcn_rcpf(1.f - b1t);
#define UR_LAZY_ELEM(W, M, V)                                   \
    {                                                           \
      const float gr = decoupled ? 0.f : a.wd * W;              \
      if (decoupled) W *= shrink;                               \
      M = a.b1 * M + c1m * gr;                                  \
      V = a.b2 * V + c2m * gr * gr;                             \
      W -= step * (M * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(V) * inv_s2 + a.eps)); \
    }
    UR_LAZY_ELEM(w.x, m.x, v.x) UR_LAZY_ELEM(w.y, m.y, v.y) UR_LAZY_ELEM(w.z, m.z, v.z) UR_LAZY_ELEM(w.w, m.w, v.w)
#undef UR_LAZY_ELEM
  }
}

// ---- the same replay without the per-element loop (Adam / AdamW, weight_decay == 0, the case above).
// The replayed weight is w_0 - m_0 S with S = sum_j c_j / (x_j + eps), x_j = sqrt(v_0) d_j: J = min(k, 192) reciprocals per ELEMENT --
// the VALU bill of a long run (a table of 100 M rows sees a row again every ~3 600 steps: every looked-up row replays the full 192
// steps, 28 K rows x 128 elements x 192 quarter-rate reciprocals = 0.17 ms per step).  Written as a series in r_j = eps / x_j,
//     1 / (x_j + eps) = (1 / x_j) (1 - r_j + r_j^2 - ...)      =>      S = (1 / sv) sum_n (-e)^n G_{n+1},
//     e = eps / sv,   sv = sqrt(v_0),   G_p = sum_j c_j d_j^-p,
// the sums G_p depend on (from, k) only -- the same for every element of the row -- so the TPR lanes of the row's group compute
// them TOGETHER (lane l takes j = l + 1, l + 1 + TPR, ...; b^j = exp2(j log2 b), no running products) and an element pays two
// reciprocals and a Horner chain.  LAZY_NT = 6 terms, used while r_j < 0.1 for every step (truncation < 1e-6 of an update that is
// itself <= lr * 10: far below the weight's fp32 resolution); elements with a smaller second moment (sqrt(v) < ~1e-7: next to no
// gradient history) take the exact loop, as do short gaps (k < LAZY_SERIES_MIN, where the loop is cheaper than the cooperative sums).
constexpr int LAZY_NT = 6;
constexpr int LAZY_SERIES_MIN = 6;
constexpr float LAZY_SERIES_R = 0.1f;
struct LazyRow {
  float G[LAZY_NT];
  float invd_max, f1, f2;
  int from, to;
  bool series;    // group-uniform: the sums are valid
};
// every lane of the row's TPR-lane group calls this with the same (from, to); t = lane index within the group
template <int TPR>
__device__ __forceinline__ LazyRow lazy_row_prepare(int from, int to, const AdamK& a, int t) {
  LazyRow r;
  r.from = from; r.to = to;
  const int k = to - from;
  // from == 0: the row was never updated (m = v = 0: nothing to replay, lazy_replay4 returns at once)
  r.series = a.algo < UR_OPT_SGD && a.wd == 0.f && k >= LAZY_SERIES_MIN && from > 0;
  if (!r.series) return r;
  const int J = min(k, LAZY_EXACT_STEPS);
  float g[LAZY_NT];
#pragma unroll
  for (int p = 0; p < LAZY_NT; ++p) g[p] = 0.f;
  float mx = 0.f;
  for (int j = t + 1; j <= J; j += TPR) {
    const float fj = (float)j, ft = (float)(from + j);
    const float p1 = __builtin_amdgcn_exp2f(fj * a.lb1), p2 = __builtin_amdgcn_exp2f(fj * a.lb2);
    const float b1t = __builtin_amdgcn_exp2f(ft * a.lb1), b2t = __builtin_amdgcn_exp2f(ft * a.lb2);
    const float c = a.lr * p1 * __builtin_amdgcn_rcpf(1.f - b1t);
    const float invd = __builtin_amdgcn_sqrtf((1.f - b2t) * __builtin_amdgcn_rcpf(p2));
    mx = fmaxf(mx, invd);
    float q = c;
#pragma unroll
    for (int p = 0; p < LAZY_NT; ++p) {
      q *= invd;
      g[p] += q;
    }
  }
#pragma unroll
  for (int off = TPR / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int p = 0; p < LAZY_NT; ++p) g[p] += __shfl_xor(g[p], off, TPR);
    mx = fmaxf(mx, __shfl_xor(mx, off, TPR));
  }
#pragma unroll
  for (int p = 0; p < LAZY_NT; ++p) r.G[p] = g[p];
  r.invd_max = mx;
  r.f1 = __builtin_amdgcn_exp2f((float)k * a.lb1);
  r.f2 = __builtin_amdgcn_exp2f((float)k * a.lb2);
  return r;
}
__device__ __forceinline__ bool lazy_series_elem(float m, float v, const LazyRow& r, float eps, float& S) {
  S = 0.f;
  if (m == 0.f) return true;                       // contributes nothing, whatever v is
  const float isv = __builtin_amdgcn_rsqf(v);      // v == 0 -> inf -> e = inf -> not < R: exact loop
  const float e = eps * isv;
  if (!(e * r.invd_max < LAZY_SERIES_R)) return false;
  float h = r.G[LAZY_NT - 1];
#pragma unroll
  for (int p = LAZY_NT - 2; p >= 0; --p) h = fmaf(-e, h, r.G[p]);
  S = isv * h;
  return true;
}
// one float4 of a row whose group prepared `r` (per lane; lanes without data simply do not call it)
__device__ __forceinline__ void lazy_row_apply(const LazyRow& r, float4& w, float4& m, float4& v, const AdamK& a) {
  if (r.to - r.from <= 0) return;
  if (r.series) {
    float sx, sy, sz, sw;
    const bool ok = lazy_series_elem(m.x, v.x, r, a.eps, sx) & lazy_series_elem(m.y, v.y, r, a.eps, sy) &
                    lazy_series_elem(m.z, v.z, r, a.eps, sz) & lazy_series_elem(m.w, v.w, r, a.eps, sw);
    if (ok) {
      w.x = fmaf(-m.x, sx, w.x); w.y = fmaf(-m.y, sy, w.y); w.z = fmaf(-m.z, sz, w.z); w.w = fmaf(-m.w, sw, w.w);
      m.x *= r.f1; m.y *= r.f1; m.z *= r.f1; m.w *= r.f1;
      v.x *= r.f2; v.y *= r.f2; v.z *= r.f2; v.w *= r.f2;
      return;
    }
  }
  lazy_replay4(w, m, v, r.from, r.to, a);
}

// (rows_reduce_kernel's epilogue: same arithmetic per element as sparse_adam_body, MODE 0)
template <int TPR>
__device__ __forceinline__ void fused_row_update(const FusedUpdate& fu, long long row, const float4 (&grad)[MAXV], int t, int d4, float scale,
                                                 float bc1, float bc2s) {
  const AdamK& a = fu.a;
  const int last = fu.last ? fu.last[row] : a.step - 1;
  LazyRow lr;
  if (fu.last) lr = lazy_row_prepare<TPR>(last, a.step - 1, a, t);   // (every lane of the group: the sums are a group effort)
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = t + k * TPR;
    if (c < d4) {
      float4 w = fu.table[row * d4 + c], m = fu.mom[row * d4 + c], v = fu.var[row * d4 + c];
      if (fu.last) lazy_row_apply(lr, w, m, v, a);
      const float4 gr = grad[k];
      opt_elem(w.x, m.x, v.x, gr.x * scale, a, bc1, bc2s);
      opt_elem(w.y, m.y, v.y, gr.y * scale, a, bc1, bc2s);
      opt_elem(w.z, m.z, v.z, gr.z * scale, a, bc1, bc2s);
      opt_elem(w.w, m.w, v.w, gr.w * scale, a, bc1, bc2s);
      fu.table[row * d4 + c] = w;
      fu.mom[row * d4 + c] = m;
      fu.var[row * d4 + c] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (fu.last && t == 0) fu.last[row] = a.step;
}

template <int TPR>
__device__ __forceinline__ void fused_row_update1(const FusedUpdate& fu, long long row, float4 gr, int last, float4 w, float4 m, float4 v,
                                                  int t, int c, bool cin, int d4, float scale, float bc1, float bc2s) {
  const AdamK& a = fu.a;
  if (fu.last) {
    const LazyRow lr = lazy_row_prepare<TPR>(last, a.step - 1, a, t);
    lazy_row_apply(lr, w, m, v, a);
  }
  opt_elem(w.x, m.x, v.x, gr.x * scale, a, bc1, bc2s);
  opt_elem(w.y, m.y, v.y, gr.y * scale, a, bc1, bc2s);
  opt_elem(w.z, m.z, v.z, gr.z * scale, a, bc1, bc2s);
  opt_elem(w.w, m.w, v.w, gr.w * scale, a, bc1, bc2s);
  if (cin) {
    fu.table[row * d4 + c] = w;
    fu.mom[row * d4 + c] = m;
    fu.var[row * d4 + c] = v;
  }
  __builtin_amdgcn_wave_barrier();
  if (fu.last && t == 0) fu.last[row] = a.step;
}

// MODE 0: update with gradient (catch-up first when last_step != null); MODE 1: catch-up only (to step-1)
template <int TPR, int MODE>
__device__ __forceinline__ void sparse_adam_body(const AdamK& a, float4* __restrict__ table, float4* __restrict__ mom,
                                                 float4* __restrict__ var, int* __restrict__ last_step,
                                                 const int* __restrict__ uniq_idx, const int* __restrict__ n_uniq_dev, long long n_max,
                                                 const float4* __restrict__ grad, int d4, const float* __restrict__ scale_dev,
                                                 int bid, int nblk, const int* __restrict__ guard_dev) {
  constexpr int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int n_uniq = (int)min((long long)*n_uniq_dev, n_max);
  const float scale = ur_step_scale(scale_dev, MODE == 1 ? nullptr : guard_dev);
  if (MODE == 0 && scale < 0.f) return;   // update guard: NaN loss or a raised id guard, the whole step is skipped (see dense_adam_kernel)
  if (bid * groups >= n_uniq) return;     // nothing for this workgroup (the grid is sized for the plan's capacity; a filtered catch-up
                                          // list is often EMPTY: 3 520 workgroups evaluating two powf for nothing were 20 us)
  const float bc1 = 1.f - powf(a.b1, (float)a.step), bc2s = sqrtf(1.f - powf(a.b2, (float)a.step));
  if (MODE != 1 && d4 <= TPR) {
    // One float4 per lane and row: FOUR rows per lane group in flight.  The rows are random 512-byte reads over tables of tens of
    // GB (w, m, v: three TLB misses per row); with one row per group the kernel is a chain of dependent round trips (plan entry ->
    // last_step -> row) at 1.4 TB/s.  Here every load of a trip is issued before the first use; indices are clamped and the loads
    // unconditional (a load behind a branch is waited for on the spot).  Same arithmetic per element as the loop below.
    constexpr int U = 4;
    const int c = min(t, d4 - 1);
    const bool cin = t < d4;
    for (int u0 = (bid * groups + g) * U; u0 < n_uniq; u0 += nblk * groups * U) {
      long long row[U];
      int last[U];
#pragma unroll
      for (int i = 0; i < U; ++i) row[i] = uniq_idx[min(u0 + i, n_uniq - 1)];
#pragma unroll
      for (int i = 0; i < U; ++i) last[i] = last_step ? last_step[row[i]] : a.step - 1;
      float4 w[U], m[U], v[U], gr[U];
#pragma unroll
      for (int i = 0; i < U; ++i) {
        w[i] = table[row[i] * d4 + c];
        m[i] = mom[row[i] * d4 + c];
        v[i] = var[row[i] * d4 + c];
        gr[i] = grad[(long long)min(u0 + i, n_uniq - 1) * d4 + c];
      }
#pragma unroll
      for (int i = 0; i < U; ++i) {
        if (u0 + i >= n_uniq || row[i] == 0) continue;   // group-uniform
        if (last_step) {
          const LazyRow lr = lazy_row_prepare<TPR>(last[i], a.step - 1, a, t);
          lazy_row_apply(lr, w[i], m[i], v[i], a);
        }
        opt_elem(w[i].x, m[i].x, v[i].x, gr[i].x * scale, a, bc1, bc2s);
        opt_elem(w[i].y, m[i].y, v[i].y, gr[i].y * scale, a, bc1, bc2s);
        opt_elem(w[i].z, m[i].z, v[i].z, gr[i].z * scale, a, bc1, bc2s);
        opt_elem(w[i].w, m[i].w, v[i].w, gr[i].w * scale, a, bc1, bc2s);
        if (cin) {
          table[row[i] * d4 + c] = w[i];
          mom[row[i] * d4 + c] = m[i];
          var[row[i] * d4 + c] = v[i];
        }
        if (last_step && t == 0) last_step[row[i]] = a.step;
      }
    }
    return;
  }
  for (int u = bid * groups + g; u < n_uniq; u += nblk * groups) {
    const long long row = uniq_idx[u];
    if (row == 0) continue;
    const int last = last_step ? last_step[row] : a.step - 1;
    if (MODE == 1 && last >= a.step - 1) continue;   // already there (updated by the step in between, or caught up ahead of time)
    if (MODE == 1 && last == 0 && a.wd == 0.f) {   // never updated: m = v = 0, every zero-gradient step is a no-op
      if (t == 0) last_step[row] = a.step - 1;
      continue;
    }
    LazyRow lr;
    if (last_step) lr = lazy_row_prepare<TPR>(last, a.step - 1, a, t);   // (every lane of the group: the sums are a group effort)
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        float4 w = table[row * d4 + c], m = mom[row * d4 + c], v = var[row * d4 + c];
        if (last_step) lazy_row_apply(lr, w, m, v, a);
        if (MODE != 1) {
          const float4 gr = grad[(long long)u * d4 + c];
          opt_elem(w.x, m.x, v.x, gr.x * scale, a, bc1, bc2s);
          opt_elem(w.y, m.y, v.y, gr.y * scale, a, bc1, bc2s);
          opt_elem(w.z, m.z, v.z, gr.z * scale, a, bc1, bc2s);
          opt_elem(w.w, m.w, v.w, gr.w * scale, a, bc1, bc2s);
        }
        table[row * d4 + c] = w;
        mom[row * d4 + c] = m;
        var[row * d4 + c] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (last_step && t == 0) last_step[row] = (MODE != 1) ? a.step : a.step - 1;
  }
}

template <int TPR, int MODE>
__global__ __launch_bounds__(256) void sparse_adam_kernel(AdamK a, float4* __restrict__ table, float4* __restrict__ mom,
                                                          float4* __restrict__ var, int* __restrict__ last_step,
                                                          const int* __restrict__ uniq_idx, const int* __restrict__ n_uniq_dev,
                                                          long long n_max, const float4* __restrict__ grad, int d4,
                                                          const float* __restrict__ scale_dev, const int* __restrict__ guard_dev) {
  if (!a.background) __builtin_amdgcn_s_setprio(3);   // (see rows_reduce_kernel)
  sparse_adam_body<TPR, MODE>(a, table, mom, var, last_step, uniq_idx, n_uniq_dev, n_max, grad, d4, scale_dev, (int)blockIdx.x, (int)gridDim.x,
                              guard_dev);
}
template <int TPR>
__global__ __launch_bounds__(256) void lazy_flush_kernel(AdamK a, float4* __restrict__ table, float4* __restrict__ mom,
                                                         float4* __restrict__ var, int* __restrict__ last_step, long long row0,
                                                         long long n, int d4) {
  constexpr int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  for (long long r = (long long)blockIdx.x * groups + g; r < n; r += (long long)gridDim.x * groups) {
    const long long row = row0 + r;
    if (row == 0) continue;
    const int last = last_step[row];
    if (last >= a.step) continue;
    const LazyRow lr = lazy_row_prepare<TPR>(last, a.step, a, t);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        float4 w = table[row * d4 + c], m = mom[row * d4 + c], v = var[row * d4 + c];
        lazy_row_apply(lr, w, m, v, a);
        table[row * d4 + c] = w;
        mom[row * d4 + c] = m;
        var[row * d4 + c] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (t == 0) last_step[row] = a.step;
  }
}

static inline int pick_tpr(int d) {
  int d4 = d / 4, t = 4;
  while (t < d4 && t < 32) t <<= 1;
  return t;
}
static inline int bits_for(long long n_rows) {
  int b = 1;
  while (b < 32 && (1LL << b) < n_rows) ++b;
  return b;
}

// ---- small batches (n <= SMALL_N = 32 768 ids: every training batch of the benchmark shapes), three launches on many CUs:
//   1. plan_chunk_sort_kernel : a workgroup per 2048-id chunk builds the keys and sorts (key, position) pairs in LDS (bitonic network
//      on the packed 64-bit value key << 11 | local position: all values distinct, so the order of equal keys IS the lookup order);
//   2. plan_chunk_rank_kernel : a thread per id; its final position = its place in its own chunk + for every OTHER chunk the number of
//      ids that precede it there (upper bound in earlier chunks, lower bound in later ones: stable) -- <= 15 independent branch-free
//      bisections over L2-resident 8 KB chunks, all in flight together; the id and its key are scattered to that position;
//   3. plan_heads_kernel : run heads -> uniq_idx / seg_start / n_uniq (/ per-owner counts).
// Replaces the one-workgroup LSD radix sort (plan_small_kernel: 4 passes of global round trips on ONE CU, 205 us at n = 28 161;
// UR_PLAN_ONEWG=1 restores it) wherever the sort is not hidden under the previous step: first step, no lookahead, evaluation,
// the row-sharded step.  Same output, bit for bit.
constexpr int MID_CHUNK = 2048;

__global__ __launch_bounds__(1024) void plan_chunk_sort_kernel(const int* __restrict__ ids_a, long long n_a, const long long* __restrict__ ids_b,
                                                               long long n_b, int W, long long n_local, unsigned* __restrict__ ckeys,
                                                               int* __restrict__ cpos, int* __restrict__ owner_counts, long long n_rows,
                                                               IdGuard gd) {
  __shared__ unsigned long long s[MID_CHUNK];
  const int tid = threadIdx.x;
  const long long n = n_a + n_b, base = (long long)blockIdx.x * MID_CHUNK;
  if (owner_counts && blockIdx.x == 0)
    for (int i = tid; i < W; i += 1024) owner_counts[i] = 0;
#pragma unroll
  for (int q = 0; q < MID_CHUNK / 1024; ++q) {
    const int i = tid + q * 1024;
    const long long g = base + i;
    unsigned key = 0xFFFFFFFFu;   // padding of the last chunk: sorts behind every real key
    if (g < n) {
      const long long id = ur_guard_id((g < n_a) ? (long long)ids_a[g] : ids_b[g - n_a], n_rows, gd);
      key = (W > 1) ? (id ? (unsigned)((id % W) * n_local + id / W + 1) : 0u) : (unsigned)id;
    }
    s[i] = ((unsigned long long)key << 11) | (unsigned)i;
  }
  __syncthreads();
  for (int k = 2; k <= MID_CHUNK; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1)), p = i | j;   // the tid-th compare-exchange pair of this stage
      const bool up = (i & k) == 0;
      const unsigned long long a = s[i], b = s[p];
      if ((a > b) == up) {
        s[i] = b;
        s[p] = a;
      }
      __syncthreads();
    }
#pragma unroll
  for (int q = 0; q < MID_CHUNK / 1024; ++q) {
    const int i = tid + q * 1024;
    ckeys[base + i] = (unsigned)(s[i] >> 11);
    cpos[base + i] = (int)(base + (long long)(s[i] & 2047u));
  }
}

// 16 lanes per id, lane q bisects chunk q: the 12 dependent loads of a bisection are the whole latency chain of the kernel (a thread per
// id with 15 chains in flight ran 120 workgroups for 38 us; this runs 1920 for the same number of loads).
__global__ __launch_bounds__(256) void plan_chunk_rank_kernel(const unsigned* __restrict__ ckeys, const int* __restrict__ cpos, int n,
                                                              int nchunks, int* __restrict__ sorted_pos, int* __restrict__ keys_sorted) {
  constexpr int MAXC = SMALL_N_MID / MID_CHUNK;   // 16 lanes per id
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int g = t / MAXC, cc = t % MAXC;          // id (chunk-sorted order), the chunk this lane searches
  const bool live = g < nchunks * MID_CHUNK;
  const int gg = live ? g : 0;
  const int orig = cpos[gg];
  const unsigned key = ckeys[gg];
  const int c = gg / MID_CHUNK;
  int cnt = 0;
  if (live && orig < n && cc < nchunks && cc != c) {
    const unsigned* a = ckeys + cc * MID_CHUNK;
    const bool le = cc < c;                       // earlier chunk: equal keys precede (stable); later chunk: they follow
    int pos = 0;
#pragma unroll
    for (int step = MID_CHUNK / 2; step >= 1; step >>= 1) {
      const unsigned v = a[pos + step - 1];
      pos += (le ? v <= key : v < key) ? step : 0;
    }
    const unsigned v = a[pos];                    // the 2048th key
    cnt = pos + ((le ? v <= key : v < key) ? 1 : 0);
  }
  cnt += __shfl_xor(cnt, 1, 64);
  cnt += __shfl_xor(cnt, 2, 64);
  cnt += __shfl_xor(cnt, 4, 64);
  cnt += __shfl_xor(cnt, 8, 64);
  if (live && orig < n && cc == 0) {
    const int rank = (gg - c * MID_CHUNK) + cnt;
    sorted_pos[rank] = orig;
    keys_sorted[rank] = (int)key;
  }
}

struct PlanWs {
  unsigned *keys0, *keys1;
  int *vals_tmp, *hist, *counts;
  long long bytes;
};
static PlanWs carve_plan(long long n, char* base) {
  PlanWs w;
  long long o = 0;
  auto take = [&](long long bytes) {
    char* p = base ? base + o : nullptr;
    o += (bytes + 255) & ~255LL;
    return p;
  };
  const long long nwaves = (n + CH - 1) / CH;
  const long long np = n + MID_CHUNK;   // (the chunk-sort path pads the last chunk with sentinel keys)
  w.keys0 = (unsigned*)take(np * 4);
  w.keys1 = (unsigned*)take(np * 4);
  w.vals_tmp = (int*)take(np * 4);
  w.hist = (int*)take(4 * nwaves * RADIX * 4);   // one [wave][digit] table per pass (<= 4 passes of 8 bits)
  w.counts = (int*)take((nwaves + 1) * 4);
  w.bytes = o;
  return w;
}


// dense[uniq_idx[u], :] += rows[u, :]   (unique rows; row 0 skipped: padding)
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const int* __restrict__ uniq_idx, const int* __restrict__ n_uniq_dev,
                                                               long long n_max, const float4* __restrict__ rows, int d4,
                                                               float4* __restrict__ dense) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = min((long long)*n_uniq_dev, n_max);
  if (i >= n * d4) return;
  const long long u = i / d4, c = i % d4;
  const long long row = uniq_idx[u];
  if (row == 0) return;
  float4 a = dense[row * d4 + c];
  const float4 b = rows[i];
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  dense[row * d4 + c] = a;
}

}  // namespace ur

using namespace ur;


// ------------------------------------------------------------------------------------------------ plan of W sorted runs
// The owner side of the row-sharded step receives, from every rank, a block of row ids that is already ascending and unique
// (each sender's plan sorted them).  The plan of their concatenation -- what ur_rows_plan would compute with a full radix
// sort (200 us in one workgroup at 28 K ids) -- is a W-way MERGE: the sorted position of an id is its offset in its own run
// plus, for every other run, the number of smaller ids (ids of EARLIER runs that are equal also come first: the order a stable
// sort gives).  One binary search per (id, run), all independent: one thread per id.
struct RunStarts { int start[66]; int W; };   // start[W] = n

__global__ __launch_bounds__(256) void plan_merge_rank_kernel(const int* __restrict__ ids, RunStarts rs, int n, int* __restrict__ sorted_pos,
                                                              int* __restrict__ keys_sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int r = 0;
  while (r + 1 < rs.W && i >= rs.start[r + 1]) ++r;
  const int x = ids[i];
  int pos = i - rs.start[r];
  for (int q = 0; q < rs.W; ++q) {
    if (q == r) continue;
    int lo = rs.start[q], hi = rs.start[q + 1];
    if (q < r) {   // upper bound: ids <= x
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids[mid] <= x) lo = mid + 1; else hi = mid; }
    } else {       // lower bound: ids < x
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids[mid] < x) lo = mid + 1; else hi = mid; }
    }
    pos += lo - rs.start[q];
  }
  sorted_pos[pos] = i;
  keys_sorted[pos] = x;
}
// heads of the runs of equal keys -> uniq_idx / seg_start / n_uniq.  Workgroup b owns positions [b * HEADS_SPAN, (b + 1) * HEADS_SPAN):
// it first COUNTS the heads in front of its span itself (coalesced loads, eight independent ones per thread in flight; the keys are a
// few hundred KB and sit in L2), then ranks its own heads with wave ballots -- one launch, no carried state between workgroups, every
// load independent of every other.  (The first version gave thread t a contiguous slice of n / 1024 positions in ONE workgroup: 2 x 28
// dependent, uncoalesced loads per thread = 69 us at C5's 28 K ids.)
constexpr int HEADS_U = 4, HEADS_SPAN = 1024 * HEADS_U;

__device__ __forceinline__ int heads_block_sum(int v, int* red, int tid) {   // sum over the 1024 threads; red: 16 ints of LDS
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  int s = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += red[w];
  return s;
}

__global__ __launch_bounds__(1024) void plan_merge_heads_kernel(const int* __restrict__ keys_sorted, int n, int* __restrict__ uniq_idx,
                                                               int* __restrict__ seg_start, int* __restrict__ n_uniq_dev,
                                                               int* __restrict__ owner_counts = nullptr, long long n_local = 1,
                                                               unsigned* __restrict__ status = nullptr) {
  __shared__ int red[16];
  __shared__ int wcnt[HEADS_U][16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int span0 = blockIdx.x * HEADS_SPAN;
  auto is_head = [&](int p) { return p < n && (p == 0 || keys_sorted[p] != keys_sorted[p - 1]); };
  // this span
  int key[HEADS_U], rank[HEADS_U];
  bool h[HEADS_U];
#pragma unroll
  for (int u = 0; u < HEADS_U; ++u) {
    const int p = span0 + u * 1024 + tid;
    key[u] = p < n ? keys_sorted[p] : 0;
    h[u] = is_head(p);
  }
  // heads in [0, span0)
  int c = 0;
  if (status) {
    // (the large-batch path, gridDim.x <= 1024: every workgroup publishes the head count of ITS span -- one word, count | ready bit, zeroed by
    // the sort's first launch -- and thread t waits for workgroup t < blockIdx.x.  The word IS the message, nothing else is ordered by it;
    // workgroups are dispatched in index order, so a predecessor is running or done.  Counting [0, span0) again in every workgroup, below,
    // is quadratic in n: 23 us at n = 153 728.)
    int mine = 0;
#pragma unroll
    for (int u = 0; u < HEADS_U; ++u) mine += h[u] ? 1 : 0;
    mine = heads_block_sum(mine, red, tid);
    if (tid == 0) __hip_atomic_store(status + blockIdx.x, 0x80000000u | (unsigned)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < (int)blockIdx.x) {
      unsigned v;
      do {
        v = __hip_atomic_load(status + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(v & 0x80000000u)) __builtin_amdgcn_s_sleep(2);
      } while (!(v & 0x80000000u));
      c = (int)(v & 0x7fffffffu);
    }
    __syncthreads();   // (red is reused)
  } else {
    for (int p0 = 0; p0 < span0; p0 += 8 * 1024) {
      bool hh[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + u * 1024 + tid;
        hh[u] = p < span0 && is_head(p);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) c += hh[u] ? 1 : 0;
    }
  }
  const int before = heads_block_sum(c, red, tid);
#pragma unroll
  for (int u = 0; u < HEADS_U; ++u) {
    const unsigned long long m = __ballot(h[u]);
    rank[u] = __popcll(m & ((1ULL << lane) - 1ULL));
    if (lane == 0) wcnt[u][wv] = __popcll(m);
  }
  __syncthreads();
  int total = before;
#pragma unroll
  for (int u = 0; u < HEADS_U; ++u) {
    int o = total;                       // heads of every earlier position group of the span ...
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int v = wcnt[u][w];
      o += w < wv ? v : 0;               // ... and of the earlier waves of this one
      total += v;
    }
    if (h[u]) {
      const int slot = o + rank[u];
      uniq_idx[slot] = key[u];
      seg_start[slot] = span0 + u * 1024 + tid;
      if (owner_counts) atomicAdd(&owner_counts[(unsigned)key[u] / n_local], 1);
    }
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    n_uniq_dev[0] = total;
    seg_start[total] = n;
  }
}
static inline int heads_grid(long long n) { return (int)((n + HEADS_SPAN - 1) / HEADS_SPAN); }

extern "C" int ur_rows_plan_merge(const int32_t* ids, int64_t n, const int32_t* host_run_start, int32_t n_runs, int32_t* uniq_idx,
                                  int32_t* seg_start, int32_t* sorted_pos, int32_t* n_uniq_dev, void* ws, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(ids && host_run_start && uniq_idx && seg_start && sorted_pos && n_uniq_dev && ws, UR_ERR_ARG, "ur_rows_plan_merge: null pointer");
  UR_REQUIRE(n > 0 && n < (1LL << 30) && n_runs >= 1 && n_runs <= 65, UR_ERR_ARG, "ur_rows_plan_merge: n=%lld runs=%d", (long long)n, n_runs);
  RunStarts rs;
  rs.W = n_runs;
  for (int q = 0; q <= n_runs; ++q) rs.start[q] = host_run_start[q];
  UR_REQUIRE(rs.start[0] == 0 && rs.start[n_runs] == n, UR_ERR_ARG, "ur_rows_plan_merge: run offsets must span [0, n]");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_SORT, st, (double)n * 4.0 * 4);
  int* keys_sorted = (int*)ws;   // n ints (the plan workspace is far larger)
  hipLaunchKernelGGL(plan_merge_rank_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, ids, rs, (int)n, sorted_pos, keys_sorted);
  UR_LAUNCH_CHECK();
  hipLaunchKernelGGL(plan_merge_heads_kernel, dim3(heads_grid(n)), dim3(1024), 0, st, keys_sorted, (int)n, uniq_idx, seg_start, n_uniq_dev);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int64_t ur_rows_plan_workspace_bytes(int64_t n) {
  if (n < 0) return UR_ERR_ARG;
  return carve_plan(n > 0 ? n : 1, nullptr).bytes;
}

static int rows_plan_impl(const int32_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b, int64_t n_rows, int W,
                          int32_t* uniq_idx, int32_t* seg_start, int32_t* sorted_pos, int32_t* n_uniq_dev,
                          int32_t* owner_counts_dev, void* ws, void* stream) {
  const long long n = n_a + n_b;
  UR_REQUIRE(n_a >= 0 && n_b >= 0 && n > 0 && n < (1LL << 31), UR_ERR_ARG, "ur_rows_plan: n_a=%lld n_b=%lld", (long long)n_a, (long long)n_b);
  UR_REQUIRE((ids_a || n_a == 0) && (ids_b || n_b == 0) && uniq_idx && seg_start && sorted_pos && n_uniq_dev && ws, UR_ERR_ARG,
             "ur_rows_plan: null pointer");
  UR_REQUIRE(n_rows > 0 && n_rows <= (1LL << 31), UR_ERR_ARG, "ur_rows_plan: n_rows=%lld", (long long)n_rows);
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_SORT, st, (double)n * 8.0);
  const IdGuard gd = id_guard();   // (every id of a training batch passes here once: the range check rides in the first pass, common.h)
  PlanWs w = carve_plan(n, (char*)ws);
  const long long n_local = (W > 1) ? (n_rows + W - 1) / W + 1 : n_rows;   // rows per shard incl. its padding row 0
  UR_REQUIRE(n_local * W <= (1LL << 31), UR_ERR_ARG, "ur_rows_plan: sharded key space too large");
  const int passes = (bits_for(W > 1 ? n_local * W : n_rows) + 7) / 8;
  // ping-pong so that the LAST pass writes vals into sorted_pos
  unsigned* kbuf[2] = {w.keys0, w.keys1};
  int* vbuf[2];
  vbuf[passes & 1] = sorted_pos;       // after `passes` swaps the result sits in index (passes & 1)
  vbuf[(passes & 1) ^ 1] = w.vals_tmp;
  static const bool no_small = ur_test_hook("plan_multi") != 0;   // test hook: the multi-launch path at every batch size
  // (two 8-bit passes -- tables of up to 65 536 rows, BASELINE config C2 -- are 5 short launches of the radix path: 23 us at n = 28 160 against
  // 35 for the chunk-sort + bisection path, whose cost does not depend on the key width; 4 passes: 40 against 35)
  if (n <= SMALL_N && !no_small && passes > 2) {
    const int nch = cdiv(n, MID_CHUNK);
    hipLaunchKernelGGL(plan_chunk_sort_kernel, dim3(nch), dim3(1024), 0, st, ids_a, (long long)n_a, (const long long*)ids_b, (long long)n_b, W,
                       n_local, w.keys0, w.vals_tmp, owner_counts_dev, (long long)n_rows, gd);
    UR_LAUNCH_CHECK();
    hipLaunchKernelGGL(plan_chunk_rank_kernel, dim3(cdiv((long long)nch * MID_CHUNK * (SMALL_N / MID_CHUNK), 256)), dim3(256), 0, st, w.keys0, w.vals_tmp, (int)n, nch,
                       sorted_pos, (int*)w.keys1);
    UR_LAUNCH_CHECK();
    hipLaunchKernelGGL(plan_merge_heads_kernel, dim3(heads_grid(n)), dim3(1024), 0, st, (const int*)w.keys1, (int)n, uniq_idx, seg_start, n_uniq_dev,
                       owner_counts_dev, n_local);
    UR_LAUNCH_CHECK();
    return UR_OK;
  }
  if (owner_counts_dev) UR_HIP(hipMemsetAsync(owner_counts_dev, 0, sizeof(int) * W, st));
  const int nchunks = cdiv(n, RCH), hgrid = heads_grid(n);
  {
    hipLaunchKernelGGL(radix_first_kernel, dim3(nchunks), dim3(256), 0, st, ids_a, (long long)n_a, (const long long*)ids_b, (long long)n_b,
                       kbuf[0], vbuf[0], W, n_local, nchunks, w.hist, passes - 1, (unsigned*)w.counts, hgrid <= 1024 ? hgrid : 0, (long long)n_rows, gd);
    UR_LAUNCH_CHECK();
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
      hipLaunchKernelGGL(radix_pass_kernel, dim3(nchunks), dim3(256), 0, st, kbuf[cur], vbuf[cur], n, 8 * p, nchunks,
                         w.hist + (long long)p * nchunks * RADIX, kbuf[cur ^ 1], vbuf[cur ^ 1],
                         p + 1 < passes ? w.hist + (long long)(p + 1) * nchunks * RADIX : (int*)nullptr);
      UR_LAUNCH_CHECK();
      cur ^= 1;
    }
    hipLaunchKernelGGL(plan_merge_heads_kernel, dim3(hgrid), dim3(1024), 0, st, (const int*)kbuf[cur], (int)n, uniq_idx, seg_start, n_uniq_dev,
                       owner_counts_dev, n_local, hgrid <= 1024 ? (unsigned*)w.counts : (unsigned*)nullptr);
    UR_LAUNCH_CHECK();
  }
  return UR_OK;
}

extern "C" int ur_rows_plan(const int32_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b, int64_t n_rows,
                            int32_t* uniq_idx, int32_t* seg_start, int32_t* sorted_pos, int32_t* n_uniq_dev, void* ws,
                            void* stream) {
  UR_TRACE_SCOPE();
  return rows_plan_impl(ids_a, n_a, ids_b, n_b, n_rows, 1, uniq_idx, seg_start, sorted_pos, n_uniq_dev, nullptr, ws, stream);
}

extern "C" int ur_rows_plan_sharded(const int32_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b, int64_t n_rows,
                                    int32_t world, int32_t* uniq_key, int32_t* seg_start, int32_t* sorted_pos,
                                    int32_t* n_uniq_dev, int32_t* owner_counts_dev, void* ws, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(world >= 1 && world <= 1024, UR_ERR_ARG, "ur_rows_plan_sharded: world=%d", world);
  return rows_plan_impl(ids_a, n_a, ids_b, n_b, n_rows, world, uniq_key, seg_start, sorted_pos, n_uniq_dev, owner_counts_dev, ws,
                        stream);
}

// idx_a[p] / idx_b[p - n_a] = u for every lookup position p in the run of unique key u: the lookups re-expressed
// as indices into the compact table of unique rows (row 0 = the padding row iff key 0 is present).
// (a thread per sorted position, its segment found by bisection over seg_start: a thread per SEGMENT leaves the padding id's
// segment -- 17 % of all positions at ML-25M-shaped data -- to one lane group, a 100-us serial walk)
__global__ __launch_bounds__(256) void compact_index_kernel(const int* __restrict__ seg_start, const int* __restrict__ sorted_pos,
                                                            const int* __restrict__ n_uniq_dev, long long n, long long n_a,
                                                            const int* __restrict__ slot, int* __restrict__ idx_a,
                                                            long long* __restrict__ idx_b) {
  const int n_uniq = *n_uniq_dev;
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= n || n_uniq <= 0 || q >= seg_start[n_uniq]) return;
  int lo = 0, hi = n_uniq;             // the u with seg_start[u] <= q < seg_start[u + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_start[mid] <= q) lo = mid; else hi = mid;
  }
  const long long p = sorted_pos[q];
  if (slot) lo = slot[lo];   // (fixed-capacity exchange: the row of the [world * cap, d] table, see ur_shard_exchange_ids)
  if (p < n_a) idx_a[p] = lo;
  else idx_b[p - n_a] = lo;
}

extern "C" int ur_compact_index(const int32_t* seg_start, const int32_t* sorted_pos, const int32_t* n_uniq_dev, int64_t n,
                                int64_t n_a, const int32_t* slot_of_uniq, int32_t* idx_a, int64_t* idx_b, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(seg_start && sorted_pos && n_uniq_dev && n > 0 && n_a >= 0 && n_a <= n, UR_ERR_ARG, "ur_compact_index: bad argument");
  UR_REQUIRE((idx_a || n_a == 0) && (idx_b || n_a == n), UR_ERR_ARG, "ur_compact_index: null output");
  UR_REQUIRE(n < (1LL << 31), UR_ERR_UNSUPPORTED, "ur_compact_index: n=%lld", (long long)n);
  const int blocks = cdiv(n, 256);
  hipLaunchKernelGGL(compact_index_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), seg_start, sorted_pos, n_uniq_dev,
                     (long long)n, (long long)n_a, slot_of_uniq, idx_a, (long long*)idx_b);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

static int rows_reduce_impl(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos,
                            const int32_t* n_uniq_dev, int64_t n, const float* rows_a, int64_t n_a, const float* coef_b,
                            const float* vec_b, int32_t G, int32_t d, float* uniq_grad, float* sumsq_dev, const int32_t* out_rows,
                            void* stream, ReduceRiders rd = ReduceRiders(), FusedUpdate fu = FusedUpdate()) {
  const int64_t n_entries = n;
  UR_REQUIRE(uniq_idx && seg_start && sorted_pos && n_uniq_dev && (uniq_grad || fu.on), UR_ERR_ARG, "ur_rows_reduce: null pointer");
  UR_REQUIRE(!(out_rows && sumsq_dev), UR_ERR_ARG, "ur_rows_reduce: out_rows with the zeroed tail");
  UR_REQUIRE(n > 0 && n_a >= 0 && n_a <= n, UR_ERR_ARG, "ur_rows_reduce: n=%lld n_a=%lld", (long long)n, (long long)n_a);
  UR_REQUIRE((rows_a || n_a == 0) && ((coef_b && vec_b && G > 0) || n_a == n), UR_ERR_ARG, "ur_rows_reduce: missing source");
  UR_REQUIRE(d > 0 && d % 4 == 0 && d <= 512, UR_ERR_ARG, "ur_rows_reduce: d=%d", d);
  hipStream_t st = as_stream(stream);
  ProfScope ps(fu.on ? PC_ADAM : PC_REDUCE, st, (double)n * d * 4.0 * (fu.on ? 7 : 2));
  const int tpr = pick_tpr(d), groups = 256 / tpr;
  int blocks = cdiv(n_entries, groups);   // (entries = the plan's capacity)
  if (fu.on && d / 4 <= tpr) blocks = cdiv(n_entries, groups * FU_ROWS);
  if (blocks > 8192) blocks = 8192;
  if (fu.on && (fu.cold_idx || fu.hot_idx)) {   // + the replay workgroups (FusedUpdate): one per CU walks the lists with a grid stride
    fu.nb_main = blocks;
    blocks += UR_BACKGROUND_BLOCKS;
  }
  const int zero_tail = sumsq_dev != nullptr;
#define GO(T, F) hipLaunchKernelGGL((rows_reduce_kernel<T, F>), dim3(blocks), dim3(256), 0, st, uniq_idx, seg_start, sorted_pos, n_uniq_dev, \
                                    (long long)n_entries, (const float4*)rows_a, (long long)n_a, coef_b, (const float4*)vec_b, G, d / 4,       \
                                    (float4*)uniq_grad, zero_tail, out_rows, rd, fu)
  if (fu.on) {
    switch (tpr) {
      case 4: GO(4, true); break;
      case 8: GO(8, true); break;
      case 16: GO(16, true); break;
      default: GO(32, true); break;
    }
  } else {
    switch (tpr) {
      case 4: GO(4, false); break;
      case 8: GO(8, false); break;
      case 16: GO(16, false); break;
      default: GO(32, false); break;
    }
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_rows_reduce(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos,
                              const int32_t* n_uniq_dev, int64_t n, const float* rows_a, int64_t n_a, const float* coef_b,
                              const float* vec_b, int32_t G, int32_t d, float* uniq_grad, float* sumsq_dev, const int32_t* out_rows,
                              void* stream) {
  UR_TRACE_SCOPE();
  return rows_reduce_impl(uniq_idx, seg_start, sorted_pos, n_uniq_dev, n, rows_a, n_a, coef_b, vec_b, G, d, uniq_grad, sumsq_dev, out_rows, stream);
}

// ur_rows_reduce of the sharded step with its riders (ReduceRiders above): step_flags_out4 != NULL: rows_a is the received gradient
// block [world * cap, d] and out4 gets the step's flags (ur_shard_step_flags); write_flag_rows != 0: this rank's flag row goes into slot 0
// of every block of uniq_grad [world * cap, d] (what ur_shard_exchange_grads does with uniq_grad == NULL).
extern "C" int ur_rows_reduce_riders(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos,
                                     const int32_t* n_uniq_dev, int64_t n, const float* rows_a, int64_t n_a, const float* coef_b,
                                     const float* vec_b, int32_t G, int32_t d, float* uniq_grad, float* sumsq_dev, const int32_t* out_rows,
                                     int32_t world, int32_t cap, float* step_flags_out4, int32_t write_flag_rows, const float* loss_out,
                                     const int32_t* flags_dev, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(world >= 1 && world <= 256 && cap > 0, UR_ERR_ARG, "ur_rows_reduce_riders: world=%d cap=%d", world, cap);
  UR_REQUIRE(!step_flags_out4 || (rows_a && n_a >= (int64_t)(world - 1) * cap + 1), UR_ERR_ARG, "ur_rows_reduce_riders: step flags need the received block as rows_a");
  UR_REQUIRE(!write_flag_rows || out_rows, UR_ERR_ARG, "ur_rows_reduce_riders: flag rows go with sums written to their slots (out_rows)");
  ReduceRiders rd;
  rd.sf_out4 = step_flags_out4; rd.fr_on = write_flag_rows ? 1 : 0; rd.fr_loss = loss_out; rd.fr_flags = flags_dev; rd.fr_guard = id_guard().dev; rd.world = world; rd.cap = cap;
  return rows_reduce_impl(uniq_idx, seg_start, sorted_pos, n_uniq_dev, n, rows_a, n_a, coef_b, vec_b, G, d, uniq_grad, sumsq_dev, out_rows, stream, rd);
}

static int launch_sparse_adam(int mode, const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step,
                              const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const float* grad, int d,
                              const float* scale, hipStream_t st) {
  ProfScope ps(PC_ADAM, st, (double)n_max * d * 4.0 * 7);
  AdamK a{cfg->lr, cfg->beta1, cfg->beta2, cfg->eps, cfg->weight_decay, cfg->step, cfg->algo};
  a.lb1 = cfg->beta1 > 0.f ? (float)log2((double)cfg->beta1) : -1e30f;
  a.lb2 = cfg->beta2 > 0.f ? (float)log2((double)cfg->beta2) : -1e30f;
  const int tpr = pick_tpr(d), groups = 256 / tpr;
  int blocks = cdiv(n_max, groups);
  if (blocks > 8192) blocks = 8192;
  // the catch-up walks a (usually short, often empty) filtered list with a grid-stride loop: a grid sized for the plan's capacity was
  // 3 520 workgroups that start, read the count and leave -- 20 us of dispatch at the tail of every step beside the dW launch
  if (mode != 0 && blocks > UR_CATCHUP_BLOCKS) blocks = UR_CATCHUP_BLOCKS;
  if (mode == 2) {   // background replay (ur_lazy_adam_catchup_background): one workgroup per CU at most, no raised priority
    a.background = 1;
    if (blocks > UR_BACKGROUND_BLOCKS) blocks = UR_BACKGROUND_BLOCKS;
  }
  if (blocks < 1) blocks = 1;
  const int* guard_dev = id_guard().dev;
#define GO(T, MD) hipLaunchKernelGGL((sparse_adam_kernel<T, MD>), dim3(blocks), dim3(256), 0, st, a, (float4*)table, (float4*)m, \
                                     (float4*)v, last_step, uniq_idx, n_uniq_dev, (long long)n_max, (const float4*)grad, d / 4, scale, guard_dev)
#define SW(MD)            \
  switch (tpr) {          \
    case 4: GO(4, MD); break;   \
    case 8: GO(8, MD); break;   \
    case 16: GO(16, MD); break; \
    default: GO(32, MD); break; \
  }
  if (mode == 0) { SW(0) } else { SW(1) }
#undef SW
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

static int check_adam(const UrAdamCfg* c, const char* who) {
  UR_REQUIRE(c != nullptr, UR_ERR_ARG, "%s: null cfg", who);
  UR_REQUIRE(c->step >= 1, UR_ERR_ARG, "%s: step=%d must be >= 1", who, c->step);
  UR_REQUIRE(c->beta1 >= 0.f && c->beta1 < 1.f && c->beta2 >= 0.f && c->beta2 < 1.f, UR_ERR_ARG, "%s: betas", who);
  return UR_OK;
}

extern "C" int ur_sparse_adam_rows(const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step,
                                   const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const float* uniq_grad,
                                   int32_t d, const float* grad_scale_dev, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_adam(cfg, "ur_sparse_adam_rows");
  if (rc) return rc;
  UR_REQUIRE(table && m && v && uniq_idx && n_uniq_dev && uniq_grad, UR_ERR_ARG, "ur_sparse_adam_rows: null pointer");
  UR_REQUIRE(d > 0 && d % 4 == 0 && d <= 512 && n_max > 0, UR_ERR_ARG, "ur_sparse_adam_rows: d=%d n_max=%lld", d, (long long)n_max);
  return launch_sparse_adam(0, cfg, table, m, v, last_step, uniq_idx, n_uniq_dev, n_max, uniq_grad, d, grad_scale_dev, as_stream(stream));
}

// ur_lazy_adam_catchup for a replay issued UNDER a step's compute (the plan stream of the step in flight): at most one workgroup per CU and
// no raised wave priority -- at its usual priority and grid the replay of 140 K rows stretched the FFN chain beside it from 63 to 152 us.
extern "C" int ur_lazy_adam_catchup_background(const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step,
                                               const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, int32_t d, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_adam(cfg, "ur_lazy_adam_catchup_background");
  if (rc) return rc;
  UR_REQUIRE(table && m && v && last_step && uniq_idx && n_uniq_dev, UR_ERR_ARG, "ur_lazy_adam_catchup_background: null pointer");
  UR_REQUIRE(d > 0 && d % 4 == 0 && d <= 512 && n_max > 0, UR_ERR_ARG, "ur_lazy_adam_catchup_background: d=%d", d);
  return launch_sparse_adam(2, cfg, table, m, v, last_step, uniq_idx, n_uniq_dev, n_max, nullptr, d, nullptr, as_stream(stream));
}

// ur_rows_reduce + ur_sparse_adam_rows in ONE launch (FusedUpdate above): the per-row gradient sums never reach HBM.  Same sums, same
// update arithmetic, bit for bit; no uniq_grad comes out, so a caller that clips by the global norm takes the two-launch path.
extern "C" int ur_rows_reduce_update(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos,
                                     const int32_t* n_uniq_dev, int64_t n, const float* rows_a, int64_t n_a, const float* coef_b,
                                     const float* vec_b, int32_t G, int32_t d, const UrAdamCfg* cfg, float* table, float* m, float* v,
                                     int32_t* last_step, const float* grad_scale_dev, const int32_t* next_cold_idx,
                                     const int32_t* next_cold_n_dev, const int32_t* next_hot_idx, const int32_t* next_hot_n_dev,
                                     int64_t next_list_max, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_adam(cfg, "ur_rows_reduce_update");
  if (rc) return rc;
  UR_REQUIRE(table && m && v, UR_ERR_ARG, "ur_rows_reduce_update: null pointer");
  UR_REQUIRE((!next_cold_idx && !next_hot_idx) || (last_step && next_list_max > 0 && (!next_cold_idx || next_cold_n_dev) &&
                                                   (!next_hot_idx || next_hot_n_dev)),
             UR_ERR_ARG, "ur_rows_reduce_update: the next batch's replay lists need last_step, their counts and a capacity");
  FusedUpdate fu;
  fu.on = 1;
  fu.a = AdamK{cfg->lr, cfg->beta1, cfg->beta2, cfg->eps, cfg->weight_decay, cfg->step, cfg->algo};
  fu.a.lb1 = cfg->beta1 > 0.f ? (float)log2((double)cfg->beta1) : -1e30f;
  fu.a.lb2 = cfg->beta2 > 0.f ? (float)log2((double)cfg->beta2) : -1e30f;
  fu.table = (float4*)table; fu.mom = (float4*)m; fu.var = (float4*)v; fu.last = last_step;
  fu.scale_dev = grad_scale_dev; fu.guard_dev = id_guard().dev;
  fu.cold_idx = next_cold_idx; fu.cold_n = next_cold_n_dev; fu.hot_idx = next_hot_idx; fu.hot_n = next_hot_n_dev; fu.list_max = next_list_max;
  return rows_reduce_impl(uniq_idx, seg_start, sorted_pos, n_uniq_dev, n, rows_a, n_a, coef_b, vec_b, G, d, nullptr, nullptr, nullptr, stream,
                          ReduceRiders(), fu);
}

// ur_rows_reduce_update on the OWNER side of the sharded step (ur_rows_reduce_riders with step_flags_out4, + the row update): rows_a is
// the received gradient block [world * cap, d]; the gradient scale is the step's flags found in slot 0 of its source blocks (1 / world, or
// skip: a NaN loss, an id out of range or a capacity overflow on ANY rank), published to step_flags_out4 as ur_shard_step_flags does.
extern "C" int ur_rows_reduce_update_owner(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos,
                                           const int32_t* n_uniq_dev, int64_t n, const float* recv_rows, int32_t d, int32_t world, int32_t cap,
                                           float* step_flags_out4, const UrAdamCfg* cfg, float* table, float* m, float* v,
                                           int32_t* last_step, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_adam(cfg, "ur_rows_reduce_update_owner");
  if (rc) return rc;
  UR_REQUIRE(table && m && v && recv_rows && step_flags_out4, UR_ERR_ARG, "ur_rows_reduce_update_owner: null pointer");
  UR_REQUIRE(world >= 1 && world <= 256 && cap > 0 && n >= (int64_t)(world - 1) * cap + 1, UR_ERR_ARG,
             "ur_rows_reduce_update_owner: world=%d cap=%d n=%lld", world, cap, (long long)n);
  FusedUpdate fu;
  fu.on = 1;
  fu.a = AdamK{cfg->lr, cfg->beta1, cfg->beta2, cfg->eps, cfg->weight_decay, cfg->step, cfg->algo};
  fu.a.lb1 = cfg->beta1 > 0.f ? (float)log2((double)cfg->beta1) : -1e30f;
  fu.a.lb2 = cfg->beta2 > 0.f ? (float)log2((double)cfg->beta2) : -1e30f;
  fu.table = (float4*)table; fu.mom = (float4*)m; fu.var = (float4*)v; fu.last = last_step;
  fu.guard_dev = id_guard().dev;
  ReduceRiders rd;
  rd.sf_out4 = step_flags_out4; rd.world = world; rd.cap = cap;
  return rows_reduce_impl(uniq_idx, seg_start, sorted_pos, n_uniq_dev, n, recv_rows, n, nullptr, nullptr, 1, d, nullptr, nullptr, nullptr, stream,
                          rd, fu);
}

extern "C" int ur_lazy_adam_catchup(const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step,
                                    const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, int32_t d, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_adam(cfg, "ur_lazy_adam_catchup");
  if (rc) return rc;
  UR_REQUIRE(table && m && v && last_step && uniq_idx && n_uniq_dev, UR_ERR_ARG, "ur_lazy_adam_catchup: null pointer");
  UR_REQUIRE(d > 0 && d % 4 == 0 && d <= 512 && n_max > 0, UR_ERR_ARG, "ur_lazy_adam_catchup: d=%d", d);
  return launch_sparse_adam(1, cfg, table, m, v, last_step, uniq_idx, n_uniq_dev, n_max, nullptr, d, nullptr, as_stream(stream));
}

// rows of a plan that have EVER been updated (last_step != 0): the only ones a catch-up has anything to do for when weight_decay == 0.
// Run next to the plan on its side stream, it takes the random last_step reads of a batch of new rows (every row of a 100 M-row table in
// a short run) off the main stream's tail.  Order of the output is arbitrary (one atomic per wave), which the catch-up does not mind.
__global__ __launch_bounds__(256) void rows_filter_touched_kernel(const int* __restrict__ uniq_idx, const int* __restrict__ n_uniq_dev,
                                                                  long long n_max, const int* __restrict__ last_step,
                                                                  int* __restrict__ out_idx, int* __restrict__ out_n) {
  const int n = (int)min((long long)*n_uniq_dev, n_max);
  const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const int row = i < n ? uniq_idx[i] : 0;
  const bool keep = i < n && row != 0 && last_step[row] != 0;
  const unsigned long long m = __ballot(keep);
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(out_n, __popcll(m));
  base = __shfl(base, 0, 64);
  if (keep) out_idx[base + __popcll(m & ((1ULL << lane) - 1ULL))] = row;
}
extern "C" int ur_rows_filter_touched(const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const int32_t* last_step,
                                      int32_t* out_idx, int32_t* out_n_dev, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(uniq_idx && n_uniq_dev && last_step && out_idx && out_n_dev && n_max > 0 && n_max < (1LL << 31), UR_ERR_ARG,
             "ur_rows_filter_touched: null pointer or n_max=%lld", (long long)n_max);
  hipStream_t st = as_stream(stream);
  UR_HIP(hipMemsetAsync(out_n_dev, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(rows_filter_touched_kernel, dim3(cdiv(n_max, 256)), dim3(256), 0, st, uniq_idx, n_uniq_dev, (long long)n_max, last_step,
                     out_idx, out_n_dev);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// The rows of a plan split against the (sorted, unique) rows of ANOTHER plan -- the step in flight's: `hot` = in both (that step updates
// them: they are current afterwards), `cold` = only in this one and, when last_step is given, ever updated before (what a catch-up
// made a step AHEAD has work for: the step in flight gives them a zero gradient).  Output order arbitrary.
__global__ __launch_bounds__(256) void rows_split_hot_kernel(const int* __restrict__ uniq_idx, const int* __restrict__ n_uniq_dev,
                                                             long long n_max, const int* __restrict__ last_step,
                                                             const int* __restrict__ excl, const int* __restrict__ excl_n_dev, int excl_max,
                                                             int* __restrict__ cold, int* __restrict__ cold_n, int* __restrict__ hot,
                                                             int* __restrict__ hot_n, int* __restrict__ hot_u, int* __restrict__ excl_mark) {
  const int n = (int)min((long long)*n_uniq_dev, n_max);
  const int ne = excl ? min(*excl_n_dev, excl_max) : 0;
  const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const int row = i < n ? uniq_idx[i] : 0;
  bool is_hot = false;
  int lo = 0;
  if (row != 0 && ne > 0) {
    int hi = ne;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (excl[mid] < row) lo = mid + 1; else hi = mid;
    }
    is_hot = lo < ne && excl[lo] == row;
  }
  const bool is_cold = row != 0 && !is_hot && (!last_step || last_step[row] != 0);
  const unsigned long long lt = (1ULL << lane) - 1ULL;
  unsigned long long m = __ballot(is_cold);
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(cold_n, __popcll(m));
  base = __shfl(base, 0, 64);
  if (is_cold) cold[base + __popcll(m & lt)] = row;
  m = __ballot(is_hot);
  base = 0;
  if (lane == 0 && m) base = atomicAdd(hot_n, __popcll(m));
  base = __shfl(base, 0, 64);
  if (is_hot) {
    const int o = base + __popcll(m & lt);
    hot[o] = row;
    if (hot_u) hot_u[o] = lo;          // its index in the other plan's unique list
    if (excl_mark) excl_mark[lo] = 1;
  }
}
extern "C" int ur_rows_split_hot(const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const int32_t* last_step,
                                 const int32_t* excl_sorted, const int32_t* excl_n_dev, int64_t excl_max, int32_t* cold_idx,
                                 int32_t* cold_n_dev, int32_t* hot_idx, int32_t* hot_n_dev, int32_t* hot_u, int32_t* excl_mark,
                                 void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(uniq_idx && n_uniq_dev && cold_idx && cold_n_dev && hot_idx && hot_n_dev && n_max > 0 && n_max < (1LL << 31), UR_ERR_ARG,
             "ur_rows_split_hot: null pointer or n_max=%lld", (long long)n_max);
  UR_REQUIRE(!excl_sorted || (excl_n_dev && excl_max > 0 && excl_max < (1LL << 31)), UR_ERR_ARG, "ur_rows_split_hot: exclusion list");
  hipStream_t st = as_stream(stream);
  UR_HIP(hipMemsetAsync(cold_n_dev, 0, sizeof(int32_t), st));
  UR_HIP(hipMemsetAsync(hot_n_dev, 0, sizeof(int32_t), st));
  if (excl_mark && excl_sorted) UR_HIP(hipMemsetAsync(excl_mark, 0, sizeof(int32_t) * (size_t)excl_max, st));
  hipLaunchKernelGGL(rows_split_hot_kernel, dim3(cdiv(n_max, 256)), dim3(256), 0, st, uniq_idx, n_uniq_dev, (long long)n_max, last_step,
                     excl_sorted, excl_n_dev, (int)excl_max, cold_idx, cold_n_dev, hot_idx, hot_n_dev, hot_u, excl_mark);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_lazy_adam_flush(const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step, int64_t row0,
                                  int64_t n, int32_t d, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(cfg != nullptr && cfg->step >= 0, UR_ERR_ARG, "ur_lazy_adam_flush: cfg");
  UR_REQUIRE(table && m && v && last_step, UR_ERR_ARG, "ur_lazy_adam_flush: null pointer");
  UR_REQUIRE(d > 0 && d % 4 == 0 && d <= 512 && n >= 0 && row0 >= 0, UR_ERR_ARG, "ur_lazy_adam_flush: d=%d", d);
  if (n == 0) return UR_OK;
  AdamK a{cfg->lr, cfg->beta1, cfg->beta2, cfg->eps, cfg->weight_decay, cfg->step, cfg->algo};
  a.lb1 = cfg->beta1 > 0.f ? (float)log2((double)cfg->beta1) : -1e30f;
  a.lb2 = cfg->beta2 > 0.f ? (float)log2((double)cfg->beta2) : -1e30f;
  const int tpr = pick_tpr(d), groups = 256 / tpr;
  long long blocks = (n + groups - 1) / groups;
  if (blocks > 16384) blocks = 16384;
  hipStream_t st = as_stream(stream);
#define GO(T) hipLaunchKernelGGL((lazy_flush_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, a, (float4*)table, (float4*)m, \
                                 (float4*)v, last_step, (long long)row0, (long long)n, d / 4)
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

extern "C" int ur_rows_scatter_add(const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const float* rows, int32_t d,
                                   float* dense, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(uniq_idx && n_uniq_dev && rows && dense && n_max > 0 && d > 0 && d % 4 == 0, UR_ERR_ARG, "ur_rows_scatter_add: bad argument");
  hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(cdiv(n_max * (d / 4), 256)), dim3(256), 0, as_stream(stream), uniq_idx, n_uniq_dev,
                     (long long)n_max, (const float4*)rows, d / 4, (float4*)dense);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
