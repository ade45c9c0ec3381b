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
                                                               long long* __restrict__ item_id, int* __restrict__ label) {
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
  const uint32_t range = (uint32_t)(n_items - 1);          // candidates are 1 + [0, range)
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
    const long long cand = 1 + (long long)r;
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

}  // namespace ur

using namespace ur;

extern "C" int ur_sample_negatives(const int64_t* user_id, const int64_t* pos_item, int32_t B, int32_t K, int64_t n_items,
                                   int64_t n_users, const int64_t* hist_ptr, const int32_t* hist_sorted, uint64_t seed,
                                   uint32_t step, int64_t* item_id, int32_t* label, void* stream) {
  UR_REQUIRE(pos_item && item_id && B > 0 && K >= 0, UR_ERR_ARG, "ur_sample_negatives: bad argument");
  UR_REQUIRE(n_items > 1 && n_items <= (1LL << 32), UR_ERR_ARG, "ur_sample_negatives: n_items=%lld", (long long)n_items);
  UR_REQUIRE(!hist_ptr || (hist_sorted && user_id), UR_ERR_ARG, "ur_sample_negatives: history needs user_id and hist_sorted");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_MISC, st, (double)B * (K + 1) * 8.0);
  const long long n = (long long)B * (K + 1);
  hipLaunchKernelGGL(sample_negatives_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const long long*)user_id,
                     (const long long*)pos_item, B, K, (long long)n_items, (long long)n_users, (const long long*)hist_ptr, hist_sorted,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), step, (long long*)item_id, label);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
