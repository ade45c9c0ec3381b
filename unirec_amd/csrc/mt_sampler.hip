// Device row builder on the REFERENCE's random stream (round 6; SURVEY.md 8 a1-a3, VERDICT r5 "missing 2").
//
// The reference draws its negatives and its history cuts from ONE process-global CPython `random` (MT19937) stream, row after row:
//   AddNegSamples.__call__   (unirec/data/transform/addnegsamples.py:90-115): per negative up to 100 x random.randint(1, n_items - 1),
//                            rejecting the row's positive and the user's history;
//   AddUserHistory.__call__  (unirec/data/transform/adduserhistory.py:32-73): 'autoregressive' with seq_last = 0 cuts the history before
//                            random.choice(n), n = the positions of the row's ids in it.
// randint / choice are Random._randbelow_with_getrandbits: k = bit_length(n) top bits of ONE 32-bit output word per try, redrawn while
// >= n.  csrc/host_sampler.cpp walks that stream on the CPU (64 K rows/s); this file walks the SAME stream on the device, so that the
// device-resident input pipeline yields the reference's ids bit for bit (tests/test_data_path.py against the reference DataLoader's own
// batches, golden G3, and against the host builder on 10^5 rows).
//
// How a sequential stream is walked in parallel.  A row's word consumption depends on where it starts, so rows form a chain -- but each
// link is cheap once the stream is indexed:
//   1. generate: a whole block of 624 outputs per twist; the twist itself is three data-parallel segments (new[i] needs old[i], old[i + 1]
//      and [i + 397 mod 624], which is OLD for i < 227 and NEW from the segment before for the rest) + the last element.
//   2. index: "is an acceptable randint word" is a property of the WORD (top `bits` bits < n_items - 1), not of its position: a prefix
//      count over the generated words gives rank[s] = acceptable words before s, and the compacted lists cval / cpos;
//   3. chain (one thread, the arrays in LDS when they fit): row r starting at word s takes candidates rank[s] .. rank[s] + K - 1 --
//      SPECULATING that none of them is the positive or in the history -- and ends behind cpos[rank[s] + K - 1]; its history cut then
//      scans on for the first word whose top bit_length(n_occ) bits are < n_occ (one or two words);
//   4. verify (all threads): every speculated candidate against the row's positive and the user's sorted history; the FIRST row with a
//      rejected candidate is replayed exactly (the reference's loop, one thread) and the chain restarts behind it.  At 100 M items a batch
//      has no such row; a 90-item test catalogue has one in every few rows and degenerates to the serial walk -- still on the device.
// The stream's state (mt[624], position) lives in device memory between batches: ur_mt_build_rows advances it by exactly the words the
// reference would have consumed.  No host synchronisation anywhere.
#include "common.h"
#include "kernels.h"

namespace ur {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_THREADS = 1024;
struct MtWs {            // carved from the caller's workspace by mt_carve
  uint32_t* words;       // [cap] tempered outputs of this batch, in stream order
  int* rank;             // [cap + 1] acceptable words before position s
  uint32_t* cval;        // [cap] top `bits` bits of the i-th acceptable word
  int* cpos;             // [cap] its position
  int* start;            // [B + 1] word offset where row r starts (chain)
  int* cidx;             // [B] rank[start[r]]
  int* nocc;             // [B] occurrences of the row's positive in the user's history (speculative path), -1 = replay the row exactly
  int cap;
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
__device__ __forceinline__ uint32_t mt_mix(uint32_t y) { return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }

// one twist of smt[624] by the whole workgroup (5 barriers).  Returns with a barrier behind the last write.
__device__ __forceinline__ void mt_twist(uint32_t* smt) {
  const int t = threadIdx.x;
  uint32_t y = 0, last_old = 0;
  if (t < MT_N - 1) y = (smt[t] & 0x80000000u) | (smt[t + 1] & 0x7fffffffu);
  if (t == MT_N - 1) last_old = smt[MT_N - 1];
  __syncthreads();
  if (t < MT_N - MT_M) smt[t] = smt[t + MT_M] ^ mt_mix(y);                      // [0, 227): partner still old
  __syncthreads();
  if (t >= MT_N - MT_M && t < 2 * (MT_N - MT_M)) smt[t] = smt[t - (MT_N - MT_M)] ^ mt_mix(y);   // [227, 454): partner new (segment 1)
  __syncthreads();
  if (t >= 2 * (MT_N - MT_M) && t < MT_N - 1) smt[t] = smt[t - (MT_N - MT_M)] ^ mt_mix(y);      // [454, 623): partner new (segment 2)
  __syncthreads();
  if (t == MT_N - 1) smt[t] = smt[MT_M - 1] ^ mt_mix((last_old & 0x80000000u) | (smt[0] & 0x7fffffffu));
  __syncthreads();
}

__device__ __forceinline__ int bit_length(unsigned long long n) { return n ? 64 - __clzll((long long)n) : 0; }

// membership in an ascending range
__device__ __forceinline__ bool in_sorted_range(const int* __restrict__ a, long long lo, long long hi, long long x) {
  const long long e = hi;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < e && a[lo] == x;
}

struct MtArgs {
  uint32_t* state;                 // [626]: mt[624], position (624 = twist first), sticky error
  const long long *user_id, *pos_item;
  int B, K;
  long long n_items, n_users;
  const long long* hist_ptr;       // nullable
  const int* hist_items;           // per-user interaction order (the cut; nullable)
  const int* hist_sorted;          // per-user ascending (membership; nullable = AddNegSamples without user2history)
  int reject_history, want_cut;    // want_cut: 'autoregressive' with seq_last = 0 (the history draw exists)
  long long* item_id;              // [B, K + 1]
  int* label;                      // [B, K + 1] nullable
  int* choice;                     // [B] nullable: index of the chosen occurrence (history order), -1 = none
  MtWs ws;
  int lds_words;                   // capacity of the LDS copies of words / rank / cpos (0: use the workspace arrays)
};

__global__ __launch_bounds__(MT_THREADS) void mt_build_rows_kernel(MtArgs a) {
  extern __shared__ uint32_t dyn[];
  __shared__ uint32_t smt[MT_N];
  __shared__ int wave_cnt[MT_THREADS / 64];
  __shared__ int sh_ngen, sh_nacc, sh_resume, sh_need, sh_bad, sh_vfrom;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.B, K = a.K, cap = a.ws.cap;
  const uint32_t range = (uint32_t)(a.n_items - 1);
  const int bits = bit_length(range);
  // the indexed stream: LDS when the batch's words fit, else the workspace
  const bool in_lds = a.lds_words >= cap;
  uint32_t* words = in_lds ? dyn : a.ws.words;
  int* rank = in_lds ? (int*)(dyn + a.lds_words) : a.ws.rank;
  int* cpos = in_lds ? (int*)(dyn + 2 * a.lds_words + 1) : a.ws.cpos;
  uint32_t* cval = a.ws.cval;   // (read by the parallel verify pass only)
  const int pos0 = (int)a.state[MT_N];
  if (t < MT_N) smt[t] = a.state[t];
  if (t == 0) { sh_ngen = 0; sh_nacc = 0; sh_resume = 0; sh_need = 0; sh_vfrom = 0; rank[0] = 0; a.ws.start[0] = 0; }
  __syncthreads();

  // ---- 1 + 2: one more block of outputs, indexed
  auto generate_block = [&]() {
    const int ngen = sh_ngen, nacc = sh_nacc;
    int lo = 0;
    if (ngen == 0 && pos0 < MT_N) lo = pos0;   // the block the stream is in the middle of
    else mt_twist(smt);
    const int n = MT_N - lo;
    if (ngen + n > cap) {                       // workspace exhausted: sticky error, the batch is garbage (host sees it at the next call)
      if (t == 0) { a.state[MT_N + 1] = 1u; sh_need = -1; }
      __syncthreads();
      return;
    }
    const bool live = t < n;
    const uint32_t w = live ? mt_temper(smt[lo + t]) : 0u;
    const uint32_t top = bits ? (w >> (32 - bits)) : 0u;
    const bool acc = live && top < range;
    const unsigned long long m = __ballot(acc);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = nacc;
    for (int q = 0; q < wave; ++q) before += wave_cnt[q];
    const int mine = before + __popcll(m & ((1ull << lane) - 1ull));
    if (live) {
      words[ngen + t] = w;
      rank[ngen + t] = mine;
      if (acc) { cval[mine] = top; cpos[mine] = ngen + t; }
    }
    __syncthreads();
    if (t == 0) {
      int tot = nacc;
      for (int q = 0; q < MT_THREADS / 64; ++q) tot += wave_cnt[q];
      sh_nacc = tot;
      sh_ngen = ngen + n;
      rank[ngen + n] = tot;
    }
    __syncthreads();
  };

  // ---- per-row facts for the speculative path
  for (int r = t; r < B; r += MT_THREADS) {
    int n_occ = 0;
    if (a.want_cut) {
      const long long u = a.user_id[r];
      const bool known = a.hist_ptr && u >= 0 && u < a.n_users && a.hist_ptr[u + 1] > a.hist_ptr[u];
      if (known) {
        if (!a.reject_history || !a.hist_sorted) n_occ = -1;   // the negatives can sit in the history too: replay the row exactly
        else {                                                  // occurrences of the positive: equal range in the sorted copy
          const long long lo0 = a.hist_ptr[u], hi0 = a.hist_ptr[u + 1], x = a.pos_item[r];
          long long lo = lo0, hi = hi0;
          while (lo < hi) { const long long mid = (lo + hi) >> 1; if (a.hist_sorted[mid] < x) lo = mid + 1; else hi = mid; }
          long long lo2 = lo, hi2 = hi0;
          while (lo2 < hi2) { const long long mid = (lo2 + hi2) >> 1; if (a.hist_sorted[mid] <= x) lo2 = mid + 1; else hi2 = mid; }
          n_occ = (int)(lo2 - lo);
        }
      }
    }
    a.ws.nocc[r] = n_occ;
    if (a.choice) a.choice[r] = -1;
  }
  // expected words of the batch, generated up front
  {
    const double p_acc = (double)range / (double)(1ull << bits);
    const int want = (int)fmin((double)cap, (double)B * ((double)K / p_acc + (a.want_cut ? 2.0 : 0.0)) * 1.05 + 64.0);
    __syncthreads();
    while (sh_ngen < want && sh_need != -1) generate_block();
  }

  // ---- the exact replay of ONE row (thread 0): the reference's loops word by word.  Returns false when it ran out of generated words
  auto replay_row = [&](int r, int s, int& end) -> bool {
    const int ngen = sh_ngen;
    const long long u = a.user_id[r], pos = a.pos_item[r];
    const bool known = a.hist_ptr && u >= 0 && u < a.n_users && a.hist_ptr[u + 1] > a.hist_ptr[u];
    const long long hb = known ? a.hist_ptr[u] : 0, he = known ? a.hist_ptr[u + 1] : 0;
    long long* out = a.item_id + (long long)r * (K + 1);
    int p = s;
    for (int k = 1; k <= K; ++k) {
      long long picked = 0;
      for (int tries = 100; tries > 0 && picked == 0; --tries) {
        uint32_t c;
        do {                                  // randint(1, n_items - 1): 1 + _randbelow(n_items - 1)
          if (p >= ngen) return false;
          c = bits ? (words[p] >> (32 - bits)) : 0u;
          ++p;
        } while (c >= range);
        const long long cand = 1 + (long long)c;
        if (cand != pos && !(a.reject_history && a.hist_sorted && known && in_sorted_range(a.hist_sorted, hb, he, cand))) picked = cand;
      }
      out[k] = picked;
    }
    if (a.want_cut && known && a.hist_items) {   // random.choice(hits), hits = history positions of any id of the group
      int hits = 0;
      for (long long i = hb; i < he; ++i) {
        const long long h = a.hist_items[i];
        bool mth = h == pos;
        if (!a.reject_history) for (int k = 1; k <= K && !mth; ++k) mth = out[k] == h;
        hits += mth;
      }
      if (hits > 0) {
        const int kb = bit_length((unsigned)hits);
        uint32_t c;
        do {
          if (p >= ngen) return false;
          c = words[p] >> (32 - kb);
          ++p;
        } while (c >= (uint32_t)hits);
        if (a.choice) a.choice[r] = (int)c;
      }
    }
    end = p;
    return true;
  };

  // ---- 3 + 4: chain, verify, replay the first rejected row, repeat
  for (;;) {
    if (sh_need == -1) break;
    __syncthreads();
    if (t == 0) {
      const int ngen = sh_ngen, nacc = sh_nacc;
      int r = sh_resume, s = a.ws.start[r];
      sh_need = 0;
      for (; r < B; ++r) {
        const int n_occ = a.ws.nocc[r];
        if (n_occ < 0) {                       // a row that cannot be speculated
          int e;
          if (!replay_row(r, s, e)) { sh_need = 1; break; }
          a.ws.cidx[r] = -1;
          s = e;
          a.ws.start[r + 1] = s;
          continue;
        }
        if (s > ngen) { sh_need = 1; break; }
        const int idx = rank[s];
        if (idx + K > nacc) { sh_need = 1; break; }
        int e = K > 0 ? cpos[idx + K - 1] + 1 : s;
        if (n_occ > 0) {
          const int kb = bit_length((unsigned)n_occ);
          uint32_t c;
          bool dry = false;
          do {
            if (e >= ngen) { dry = true; break; }
            c = words[e] >> (32 - kb);
            ++e;
          } while (c >= (uint32_t)n_occ);
          if (dry) { sh_need = 1; break; }
          if (a.choice) a.choice[r] = (int)c;
        }
        a.ws.cidx[r] = idx;
        s = e;
        a.ws.start[r + 1] = s;
      }
      sh_resume = r;     // rows < r are chained
      sh_bad = B;
    }
    __syncthreads();
    const int chained = sh_resume;
    // verify the speculated rows (those with cidx >= 0 among the chained ones that are not final yet) and write their ids
    {
      const long long total = (long long)chained * (K + 1);
      for (long long q = (long long)sh_vfrom * (K + 1) + t; q < total; q += MT_THREADS) {
        const int r = (int)(q / (K + 1)), k = (int)(q % (K + 1));
        const int idx = a.ws.cidx[r];
        const long long pos = a.pos_item[r];
        if (a.label) a.label[q] = (k == 0);
        if (k == 0) { a.item_id[q] = pos; continue; }
        if (idx < 0) continue;                 // replayed exactly: already written
        const long long cand = 1 + (long long)cval[idx + k - 1];
        bool ok = cand != pos;
        if (ok && a.reject_history && a.hist_sorted) {
          const long long u = a.user_id[r];
          if (a.hist_ptr && u >= 0 && u < a.n_users) ok = !in_sorted_range(a.hist_sorted, a.hist_ptr[u], a.hist_ptr[u + 1], cand);
        }
        a.item_id[q] = cand;
        if (!ok) atomicMin(&sh_bad, r);
      }
    }
    __syncthreads();
    const int bad = sh_bad;
    if (bad < chained) {                       // replay that row exactly; everything behind it is re-chained
      // Parallel replay first.  Whether a candidate is rejected depends on its VALUE only (the positive, the user's history), so the row's
      // negatives are the first K valid ones among candidates cidx, cidx + 1, ...: flag a window of K + 96 of them with all threads, count,
      // and let the valid candidate number k take slot k.  The 100-tries cap of a slot cannot bind while the window holds < 100 rejected
      // candidates in all; otherwise -- or when the window runs past the generated words -- the serial walk below does it.
      bool done_par = false;
      {
        const int idx0 = a.ws.cidx[bad], n_occ = a.ws.nocc[bad];
        const int win = K + 96;
        if (idx0 >= 0 && n_occ >= 0 && idx0 + win <= sh_nacc) {        // (uniform: shared values behind a barrier)
          const long long u = a.user_id[bad], pos = a.pos_item[bad];
          const bool hist = a.reject_history && a.hist_sorted && a.hist_ptr && u >= 0 && u < a.n_users;
          const long long hb = hist ? a.hist_ptr[u] : 0, he = hist ? a.hist_ptr[u + 1] : 0;
          long long* out = a.item_id + (long long)bad * (K + 1);
          int base = 0;                                                  // valid candidates in front of the chunk
          if (t == 0) { sh_bad = -1; }                                   // (re-used: candidate index of the K-th valid one)
          __syncthreads();
          for (int c0 = 0; c0 < win && base < K; c0 += MT_THREADS) {
            const int j = c0 + t;
            bool ok = false;
            long long cand = 0;
            if (j < win) {
              cand = 1 + (long long)cval[idx0 + j];
              ok = cand != pos && !(hist && in_sorted_range(a.hist_sorted, hb, he, cand));
            }
            const unsigned long long m = __ballot(ok);
            if (lane == 0) wave_cnt[wave] = __popcll(m);
            __syncthreads();
            int before = base;
            for (int q = 0; q < wave; ++q) before += wave_cnt[q];
            const int ord = before + __popcll(m & ((1ull << lane) - 1ull));
            if (ok && ord < K) {
              out[1 + ord] = cand;
              if (ord == K - 1) sh_bad = j;
            }
            int tot = base;
            for (int q = 0; q < MT_THREADS / 64; ++q) tot += wave_cnt[q];
            __syncthreads();
            base = tot;
          }
          const int jl = sh_bad;                                         // window index of the last candidate taken
          if (K == 0 || (jl >= 0 && jl + 1 - K < 100)) {                 // fewer than 100 rejected candidates were walked over
            if (t == 0) {
              int e = K > 0 ? cpos[idx0 + jl] + 1 : a.ws.start[bad];
              bool dry = false;
              if (n_occ > 0) {
                const int kb = bit_length((unsigned)n_occ), ngen = sh_ngen;
                uint32_t c;
                do {
                  if (e >= ngen) { dry = true; break; }
                  c = words[e] >> (32 - kb);
                  ++e;
                } while (c >= (uint32_t)n_occ);
                if (!dry && a.choice) a.choice[bad] = (int)c;
              }
              if (!dry) {
                a.ws.cidx[bad] = -1;
                a.ws.nocc[bad] = -2;
                a.ws.start[bad + 1] = e;
                sh_resume = bad + 1;
                sh_vfrom = bad + 1;
                sh_need = 0;
                sh_bad = -7;                                             // "done in parallel"
              }
            }
            __syncthreads();
            done_par = sh_bad == -7;
          }
          __syncthreads();
        }
      }
      if (done_par) continue;
      if (t == 0) {
        int e;
        if (replay_row(bad, a.ws.start[bad], e)) {
          a.ws.cidx[bad] = -1;
          a.ws.nocc[bad] = -2;                 // (final: the chain skips it from now on)
          a.ws.start[bad + 1] = e;
          sh_resume = bad + 1;
          sh_vfrom = bad + 1;                  // rows up to and including this one are final
          sh_need = 0;
        } else {
          sh_resume = bad;
          sh_vfrom = bad;
          a.ws.nocc[bad] = -1;                 // replayed by the chain once more words exist
          sh_need = 1;
        }
      }
      __syncthreads();
      if (sh_need == 1) { generate_block(); generate_block(); }
      continue;
    }
    if (chained >= B) break;                   // every row chained and verified
    if (sh_need == 1) { generate_block(); generate_block(); }
  }
  __syncthreads();
  // ---- the stream's new state: the initial state advanced by the words consumed
  const int consumed = a.ws.start[B];
  if (t < MT_N) smt[t] = a.state[t];
  __syncthreads();
  int left = consumed, pos = pos0;
  if (pos < MT_N) {
    const int n = MT_N - pos;
    if (left <= n) { pos += left; left = 0; } else { left -= n; pos = MT_N; }
  }
  while (left > 0) {                           // (uniform over the workgroup: every thread computes the same counts)
    mt_twist(smt);
    if (left <= MT_N) { pos = left; left = 0; } else left -= MT_N;
  }
  __syncthreads();
  if (sh_need != -1) {
    if (t < MT_N) a.state[t] = smt[t];
    if (t == 0) a.state[MT_N] = (uint32_t)pos;
  }
}

static size_t mt_ws_bytes(int B, int K, long long n_items, int want_cut, int* cap_out) {
  const uint32_t range = (uint32_t)(n_items - 1);
  int bits = 0;
  for (uint32_t r = range; r; r >>= 1) ++bits;
  const double p_acc = (double)range / (double)(1ull << bits);
  // twice the expected words + room for validity rejections; capacity is a bound, not a cost (only generated words are touched)
  double words = (double)B * ((double)K / p_acc + (want_cut ? 2.0 : 0.0)) * 2.0 + 16.0 * MT_N;
  if (words > 2.0e8) words = 2.0e8;
  const int cap = (int)((((long long)words + MT_N - 1) / MT_N) * MT_N);
  if (cap_out) *cap_out = cap;
  return (size_t)cap * 4 * 4 + 64 + (size_t)(3 * B + 8) * 4;
}

}  // namespace ur

using namespace ur;

extern "C" int64_t ur_mt_workspace_bytes(int32_t B, int32_t K, int64_t n_items, int32_t want_cut) {
  if (B <= 0 || K < 0 || n_items < 2) return -1;
  return (int64_t)mt_ws_bytes(B, K, n_items, want_cut, nullptr);
}

extern "C" int ur_mt_build_rows(uint32_t* state, const int64_t* user_id, const int64_t* pos_item, int32_t B, int32_t K, int64_t n_items,
                                int64_t n_users, const int64_t* hist_ptr, const int32_t* hist_items, const int32_t* hist_sorted,
                                int32_t reject_history, int32_t want_cut, int64_t* item_id, int32_t* label, int32_t* choice, void* ws,
                                void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(state && pos_item && item_id && ws && B > 0 && K >= 0, UR_ERR_ARG, "ur_mt_build_rows: bad argument");
  UR_REQUIRE(n_items > 1 && n_items <= (1LL << 32), UR_ERR_ARG, "ur_mt_build_rows: n_items=%lld", (long long)n_items);
  UR_REQUIRE(!reject_history || (hist_ptr && hist_sorted && user_id), UR_ERR_ARG, "ur_mt_build_rows: history rejection needs user_id, hist_ptr, hist_sorted");
  UR_REQUIRE(!want_cut || (hist_ptr && hist_items && user_id && choice), UR_ERR_ARG, "ur_mt_build_rows: the history cut needs user_id, hist_ptr, hist_items, choice");
  hipStream_t st = as_stream(stream);
  int cap = 0;
  (void)mt_ws_bytes(B, K, n_items, want_cut, &cap);
  MtArgs a{};
  a.state = state; a.user_id = (const long long*)user_id; a.pos_item = (const long long*)pos_item; a.B = B; a.K = K;
  a.n_items = n_items; a.n_users = n_users; a.hist_ptr = (const long long*)hist_ptr; a.hist_items = hist_items; a.hist_sorted = hist_sorted;
  a.reject_history = reject_history != 0; a.want_cut = want_cut != 0;
  a.item_id = (long long*)item_id; a.label = label; a.choice = choice;
  char* p = (char*)ws;
  a.ws.cap = cap;
  a.ws.words = (uint32_t*)p; p += (size_t)cap * 4;
  a.ws.rank = (int*)p; p += (size_t)cap * 4 + 16;
  a.ws.cval = (uint32_t*)p; p += (size_t)cap * 4;
  a.ws.cpos = (int*)p; p += (size_t)cap * 4;
  a.ws.start = (int*)p; p += (size_t)(B + 2) * 4;
  a.ws.cidx = (int*)p; p += (size_t)(B + 2) * 4;
  a.ws.nocc = (int*)p;
  // words / rank / cpos in LDS when the batch's capacity fits 144 KB
  size_t lds = 0;
  if ((size_t)cap * 12 + 64 <= 144 * 1024) { a.lds_words = cap + 4; lds = (size_t)a.lds_words * 12 + 64; }
  static const hipError_t attr = hipFuncSetAttribute((const void*)mt_build_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);
  (void)attr;
  ProfScope ps(PC_MISC, st, (double)B * (K + 1) * 8.0);
  hipLaunchKernelGGL(mt_build_rows_kernel, dim3(1), dim3(MT_THREADS), lds, st, a);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
