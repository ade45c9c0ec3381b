// Device-side negative sampler: the reference's rule -- uniform over [1, N-1], reject the
// row's positive and every item of the user's history, at most 100 tries, else id 0
// (unirec/data/transform/addnegsamples.py:67-80,97-108) -- on a counter-based generator so that every
// (row, slot, try) is independent of execution order: Philox4x32-10 keyed by the seed, counter =
// (step, row, slot, try).  Bit-exact against oracle/philox_ref.py; NOT the reference's MT19937 stream (that one is
// reproduced on the host by host_sampler.cpp).
#include "common.h"
#include "kernels.h"

namespace ur {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one thread per (row, slot).  hist_sorted ranges are ascending (binary search).
__global__ __launch_bounds__(256) void sample_negatives_kernel(const long long* __restrict__ user_id, const long long* __restrict__ pos_item,
                                                               int B, int K, long long n_items, long long n_users,
                                                               const long long* __restrict__ hist_ptr, const int* __restrict__ hist_sorted,
                                                               uint32_t seed_lo, uint32_t seed_hi, uint32_t step,
                                                               long long* __restrict__ item_id, int* __restrict__ label,
                                                               const double* __restrict__ alias_odds, const long long* __restrict__ alias_idx) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * (K + 1)) return;
  const int b = t / (K + 1), k = t % (K + 1);
  const long long pos = pos_item[b];
  if (label) label[t] = (k == 0);
  if (k == 0) {
    item_id[t] = pos;
    return;
  }
  const long long u = user_id ? user_id[b] : -1;
  const bool known = hist_ptr && u >= 0 && u < n_users;
  const long long hb = known ? hist_ptr[u] : 0, he = known ? hist_ptr[u + 1] : 0;
  const uint32_t range = (uint32_t)(n_items - 1);          // uniform draw: candidates are 1 + [0, range)
  int bits = 0;
  for (uint32_t r = range; r; r >>= 1) ++bits;              // CPython-style: top `bits` bits, reject >= range
  long long picked = 0;
  for (uint32_t tr = 0; tr < 100u && picked == 0; ++tr) {
    uint32_t w[4];
    philox4x32_10(step, (uint32_t)b, (uint32_t)(k - 1), tr, seed_lo, seed_hi, w);
    uint32_t r = range;                                      // invalid
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = w[q] >> (32 - bits);
      if (r >= range && c < range) r = c;
    }
    if (r >= range) r = (uint32_t)(((unsigned long long)w[3] * range) >> 32);   // p < 2^-4 per try: multiply-shift fallback
    long long cand = 1 + (long long)r;
    if (alias_odds) {
      // popularity-biased draw (unirec/utils/sampling.py:26-30): x = random() * N; i = int(x); alias[i] if x - i > odds[i] else i,
      // with random() built from two words the way CPython does it: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53
      const double u = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) * (1.0 / 9007199254740992.0);
      const double x = u * (double)n_items;
      const long long i = (long long)x;
      cand = (x - (double)i) > alias_odds[i] ? alias_idx[i] : i;
      if (cand <= 0) continue;   // item 0 (weight 0) can only come out with x - i == 0: not a candidate
    }
    bool ok = cand != pos;
    if (ok && he > hb) {                                     // binary search in the user's sorted history
      long long lo = hb, hi = he;
      while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (hist_sorted[mid] < cand) lo = mid + 1; else hi = mid;
      }
      ok = !(lo < he && hist_sorted[lo] == cand);
    }
    if (ok) picked = cand;
  }
  item_id[t] = picked;
}


// ------------------------------------------------------------------------------------------------ history rows
// item_seq[b,:] for a batch, on the device: AddUserHistory (unirec/data/transform/adduserhistory.py:
// 32-73) + the left padding of SeqRecDataset (unirec/data/dataset/seqrecdataset.py:60-68) over a CSR history in HBM.
//   'unorder'        : history items that occur in the row's id group are zeroed;
//   'autoregressive' : the history is cut before one occurrence of an id of the group -- the last one (seq_last) or a
//                      uniformly chosen one: index = (philox(step, row, 0xFFFFFFFF, 0)[0] * count) >> 32;
//   then the last L items, left-padded with 0.  A user without history yields the reference's [0] (length 1).
// One wave per row.  match_all = 0: only the positive (column 0) can occur in the history (negatives were sampled with
// history rejection); 1: every id of the group is compared.
__global__ __launch_bounds__(256) void build_seq_kernel(const long long* __restrict__ user_id, const long long* __restrict__ item_id,
                                                        int B, int G, long long n_users, const long long* __restrict__ hist_ptr,
                                                        const int* __restrict__ hist_items, int mask_mode, int seq_last, int match_all,
                                                        int L, uint32_t seed_lo, uint32_t seed_hi, uint32_t step,
                                                        int* __restrict__ item_seq, long long* __restrict__ seq_len,
                                                        const int* __restrict__ choice) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;   // wave-uniform
  const long long u = user_id[b];
  const bool known = u >= 0 && u < n_users;
  const long long hb = known ? hist_ptr[u] : 0, he = known ? hist_ptr[u + 1] : 0;
  const long long len = he - hb;
  const long long* ids = item_id + (long long)b * G;
  const long long pos = ids[0];
  auto match = [&](int h) -> bool {
    if (!match_all) return (long long)h == pos;
    bool m = false;
    for (int g = 0; g < G; ++g) m |= ids[g] == (long long)h;
    return m;
  };
  long long keep = len;   // history[:keep] survives
  if (mask_mode == 1 && len > 0) {
    int count = 0;
    for (long long c = 0; c < len; c += 64) {
      const long long i = c + lane;
      count += __popcll(__ballot(i < len && match(hist_items[hb + i])));
    }
    if (count > 0) {
      int t = count - 1;
      if (!seq_last && choice) {
        t = min(max(choice[b], 0), count - 1);   // the occurrence the MT19937 stream chose (ur_mt_build_rows): random.choice(n)
      } else if (!seq_last) {
        uint32_t w[4];
        philox4x32_10(step, (uint32_t)b, 0xFFFFFFFFu, 0u, seed_lo, seed_hi, w);
        t = (int)(((unsigned long long)w[0] * (unsigned)count) >> 32);
      }
      int run = 0;
      for (long long c = 0; c < len; c += 64) {   // position of the t-th occurrence
        const long long i = c + lane;
        unsigned long long m = __ballot(i < len && match(hist_items[hb + i]));
        const int n = __popcll(m);
        if (run + n > t) {
          for (int k = t - run; k > 0; --k) m &= m - 1;
          keep = c + (__ffsll((long long)m) - 1);
          break;
        }
        run += n;
      }
    }
  }
  for (int j = lane; j < L; j += 64) {
    const long long src = keep - L + j;
    int v = src >= 0 ? hist_items[hb + src] : 0;
    if (mask_mode == 0 && src >= 0 && match(v)) v = 0;
    item_seq[(long long)b * L + j] = v;
  }
  if (lane == 0 && seq_len) seq_len[b] = len == 0 ? (L < 1 ? L : 1) : (keep < L ? keep : L);
}

}  // namespace ur

using namespace ur;

static int sample_negatives_impl(const int64_t* user_id, const int64_t* pos_item, int32_t B, int32_t K, int64_t n_items, int64_t n_users,
                                 const int64_t* hist_ptr, const int32_t* hist_sorted, const double* alias_odds, const int64_t* alias_idx,
                                 uint64_t seed, uint32_t step, int64_t* item_id, int32_t* label, void* stream) {
  UR_REQUIRE(pos_item && item_id && B > 0 && K >= 0, UR_ERR_ARG, "ur_sample_negatives: bad argument");
  UR_REQUIRE(n_items > 1 && n_items <= (1LL << 32), UR_ERR_ARG, "ur_sample_negatives: n_items=%lld", (long long)n_items);
  UR_REQUIRE(!hist_ptr || (hist_sorted && user_id), UR_ERR_ARG, "ur_sample_negatives: history needs user_id and hist_sorted");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_MISC, st, (double)B * (K + 1) * 8.0);
  const long long n = (long long)B * (K + 1);
  hipLaunchKernelGGL(sample_negatives_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const long long*)user_id,
                     (const long long*)pos_item, B, K, (long long)n_items, (long long)n_users, (const long long*)hist_ptr, hist_sorted,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), step, (long long*)item_id, label, alias_odds,
                     (const long long*)alias_idx);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_sample_negatives(const int64_t* user_id, const int64_t* pos_item, int32_t B, int32_t K, int64_t n_items,
                                   int64_t n_users, const int64_t* hist_ptr, const int32_t* hist_sorted, uint64_t seed,
                                   uint32_t step, int64_t* item_id, int32_t* label, void* stream) {
  UR_TRACE_SCOPE();
  return sample_negatives_impl(user_id, pos_item, B, K, n_items, n_users, hist_ptr, hist_sorted, nullptr, nullptr, seed, step, item_id, label,
                               stream);
}

extern "C" int ur_sample_negatives_pop(const int64_t* user_id, const int64_t* pos_item, int32_t B, int32_t K, int64_t n_items,
                                       int64_t n_users, const int64_t* hist_ptr, const int32_t* hist_sorted, const double* alias_odds,
                                       const int64_t* alias_idx, uint64_t seed, uint32_t step, int64_t* item_id, int32_t* label,
                                       void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(alias_odds && alias_idx, UR_ERR_ARG, "ur_sample_negatives_pop: null alias table");
  return sample_negatives_impl(user_id, pos_item, B, K, n_items, n_users, hist_ptr, hist_sorted, alias_odds, alias_idx, seed, step, item_id,
                               label, stream);
}

extern "C" int ur_device_build_seq(const int64_t* user_id, const int64_t* item_id, int32_t B, int32_t G, int64_t n_users,
                                   const int64_t* hist_ptr, const int32_t* hist_items, int32_t mask_mode, int32_t seq_last,
                                   int32_t match_all, int32_t L, uint64_t seed, uint32_t step, int32_t* item_seq, int64_t* seq_len,
                                   void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(user_id && item_id && hist_ptr && hist_items && item_seq, UR_ERR_ARG, "ur_device_build_seq: null pointer");
  UR_REQUIRE(B > 0 && G > 0 && L > 0 && n_users >= 0, UR_ERR_ARG, "ur_device_build_seq: B=%d G=%d L=%d", B, G, L);
  UR_REQUIRE(mask_mode >= 0 && mask_mode <= 2, UR_ERR_ARG, "ur_device_build_seq: mask_mode=%d", mask_mode);
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_MISC, st, (double)B * L * 4.0);
  hipLaunchKernelGGL(build_seq_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, (const long long*)user_id, (const long long*)item_id, B, G,
                     (long long)n_users, (const long long*)hist_ptr, hist_items, mask_mode, seq_last, match_all, L,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), step, item_seq, (long long*)seq_len, (const int*)nullptr);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// the same rows with the cut chosen by the caller: choice[b] = index (history order) of the occurrence the history is cut before -- what
// ur_mt_build_rows drew from the reference's MT19937 stream for row b
extern "C" int ur_device_build_seq_choice(const int64_t* user_id, const int64_t* item_id, int32_t B, int32_t G, int64_t n_users,
                                          const int64_t* hist_ptr, const int32_t* hist_items, int32_t mask_mode, int32_t seq_last,
                                          int32_t match_all, int32_t L, const int32_t* choice, int32_t* item_seq, int64_t* seq_len,
                                          void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(user_id && item_id && hist_ptr && hist_items && item_seq && choice, UR_ERR_ARG, "ur_device_build_seq_choice: null pointer");
  UR_REQUIRE(B > 0 && G > 0 && L > 0 && n_users >= 0, UR_ERR_ARG, "ur_device_build_seq_choice: B=%d G=%d L=%d", B, G, L);
  UR_REQUIRE(mask_mode >= 0 && mask_mode <= 2, UR_ERR_ARG, "ur_device_build_seq_choice: mask_mode=%d", mask_mode);
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_MISC, st, (double)B * L * 4.0);
  hipLaunchKernelGGL(build_seq_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, (const long long*)user_id, (const long long*)item_id, B, G,
                     (long long)n_users, (const long long*)hist_ptr, hist_items, mask_mode, seq_last, match_all, L, 0u, 0u, 0u, item_seq,
                     (long long*)seq_len, choice);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
