// Host-side (CPU) data path of the training rows, native instead of per-sample Python:
//   * MT19937 with CPython's seeding and derived methods, so that a stream seeded like `random.seed(s)` yields
//     the SAME negative ids / history cuts as the reference's AddNegSamples / AddUserHistory when they run on one
//     stream (num_workers=0)  -- unirec/data/transform/addnegsamples.py:67-115, adduserhistory.py:32-73,
//     unirec/utils/sampling.py:9-31, unirec/data/dataset/seqrecdataset.py:38-68;
//   * ur_host_build_rows: negatives + history truncation + left padding for a whole batch of (user, item) rows.
// Host pointers only; no HIP.  (The reference runs this per sample inside Dataset.__getitem__: 13 us/row at K=4,
// 0.8 ms/row at K=1000 -- SURVEY.md section 6.)
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

struct MT {
  uint32_t mt[624];
  int idx;
  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void init_by_array(const uint32_t* key, int len) {   // CPython random.seed(int): key = 32-bit LE chunks of |seed|
    init_genrand(19650218u);
    int i = 1, j = 0;
    for (int k = std::max(624, len); k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
      if (++j >= len) j = 0;
    }
    for (int k = 623; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  uint64_t getrandbits(int k) {   // k <= 64
    if (k <= 32) return next() >> (32 - k);
    const uint64_t lo = next();
    const uint64_t hi = next() >> (64 - k);
    return lo | (hi << 32);
  }
  uint64_t randbelow(uint64_t n) {   // Lib/random.py _randbelow_with_getrandbits
    int k = 0;
    for (uint64_t t = n; t; t >>= 1) ++k;
    uint64_t r = getrandbits(k);
    while (r >= n) r = getrandbits(k);
    return r;
  }
  double random() {
    const uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
};

struct Sampler {
  MT rng;
  std::vector<double> odds;     // alias table (popularity sampling); empty = uniform
  std::vector<int64_t> alias;
};

inline bool in_sorted(const int32_t* b, const int32_t* e, int64_t x) { return std::binary_search(b, e, (int32_t)x); }

}  // namespace

extern "C" void* ur_host_sampler_create(uint64_t seed) {
  Sampler* s = new Sampler();
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  s->rng.init_by_array(key, key[1] ? 2 : 1);
  return s;
}
extern "C" void ur_host_sampler_destroy(void* h) { delete (Sampler*)h; }
extern "C" uint64_t ur_host_sampler_getrandbits(void* h, int k) { return ((Sampler*)h)->rng.getrandbits(k); }
extern "C" double ur_host_sampler_random(void* h) { return ((Sampler*)h)->rng.random(); }
extern "C" int ur_host_sampler_state(void* h, uint32_t* out625) {
  UR_REQUIRE(h && out625, UR_ERR_ARG, "ur_host_sampler_state: null pointer");
  const MT& r = ((Sampler*)h)->rng;
  memcpy(out625, r.mt, sizeof(r.mt));
  out625[624] = (uint32_t)r.idx;
  return UR_OK;
}
extern "C" int64_t ur_host_sampler_randint(void* h, int64_t a, int64_t b) { return a + (int64_t)((Sampler*)h)->rng.randbelow((uint64_t)(b - a + 1)); }

// Alias table as unirec/utils/sampling.py:9-24 builds it (same traversal order => same table): odds[i] in [0,1], alias[i]
// (-1 where the reference keeps (1, None)).  Host pointers.
extern "C" int ur_alias_table_build(const double* w, int64_t n, double* odds, int64_t* alias) {
  UR_REQUIRE(w && odds && alias && n > 0, UR_ERR_ARG, "ur_alias_table_build: bad argument");
  double sum = 0;
  for (int64_t i = 0; i < n; ++i) sum += w[i];
  const double avg = sum / (double)n;
  for (int64_t i = 0; i < n; ++i) { odds[i] = 1.0; alias[i] = -1; }
  int64_t si = 0, bi = 0;   // generators over smalls (w < avg) and bigs (w >= avg), both in index order
  auto next_small = [&](int64_t from) { while (from < n && !(w[from] < avg)) ++from; return from; };
  auto next_big = [&](int64_t from) { while (from < n && !(w[from] >= avg)) ++from; return from; };
  si = next_small(0);
  bi = next_big(0);
  int64_t small_i = si < n ? si : -1, big_i = bi < n ? bi : -1;
  double small_v = small_i >= 0 ? w[small_i] / avg : 0, big_v = big_i >= 0 ? w[big_i] / avg : 0;
  while (big_i >= 0 && small_i >= 0) {
    odds[small_i] = small_v;
    alias[small_i] = big_i;
    big_v = big_v - (1.0 - small_v);
    if (big_v < 1.0) {
      small_i = big_i; small_v = big_v;
      bi = next_big(bi + 1);
      big_i = bi < n ? bi : -1;
      big_v = big_i >= 0 ? w[big_i] / avg : 0;
    } else {
      si = next_small(si + 1);
      small_i = si < n ? si : -1;
      small_v = small_i >= 0 ? w[small_i] / avg : 0;
    }
  }
  return UR_OK;
}

// popularity-biased negatives: weights w[i] (already pop^alpha / sum with w[0] = 0)
extern "C" int ur_host_sampler_set_alias(void* h, const double* w, int64_t n) {
  UR_REQUIRE(h && w && n > 0, UR_ERR_ARG, "ur_host_sampler_set_alias: bad argument");
  Sampler* s = (Sampler*)h;
  s->odds.assign(n, 1.0);
  s->alias.assign(n, -1);
  return ur_alias_table_build(w, n, s->odds.data(), s->alias.data());
}

static inline int64_t draw_one(Sampler* s, int64_t n_items) {
  if (s->odds.empty()) return 1 + (int64_t)s->rng.randbelow((uint64_t)(n_items - 1));   // random.randint(1, n_items-1)
  const double r = s->rng.random() * (double)s->odds.size();
  const int64_t i = (int64_t)r;
  return (r - (double)i) > s->odds[i] ? s->alias[i] : i;
}

// Rows of one batch.  hist_ptr[n_users+1] / hist_items: each user's history in interaction order;
// hist_sorted: the same per-user ranges sorted ascending (membership tests).  A user id >= n_users or an empty
// range = unknown user (history [0], adduserhistory.py:19,41-43).  mask_mode: 0 = unorder, 1 = autoregressive,
// 2 = anything else (history unchanged).  reject_history: AddNegSamples was given user2history.
// Outputs: item_id int64[n, n_neg+1] (positive first), item_seq int32[n, L] (left-padded), seq_len int64[n].
extern "C" int ur_host_build_rows(void* h, const int64_t* user_id, const int64_t* pos_item, int64_t n, int64_t n_users,
                                  int64_t n_items, int32_t n_neg, const int64_t* hist_ptr, const int32_t* hist_items,
                                  const int32_t* hist_sorted, int32_t reject_history, int32_t mask_mode, int32_t seq_last,
                                  int32_t L, int64_t* item_id, int32_t* item_seq, int64_t* seq_len) {
  UR_REQUIRE(h && user_id && pos_item && item_id && n >= 0 && n_items > 1 && n_neg >= 0, UR_ERR_ARG, "ur_host_build_rows: bad argument");
  UR_REQUIRE(!item_seq || (hist_ptr && hist_items && L > 0 && seq_len), UR_ERR_ARG, "ur_host_build_rows: history arrays required for item_seq");
  UR_REQUIRE(!reject_history || (hist_ptr && hist_sorted), UR_ERR_ARG, "ur_host_build_rows: hist_sorted required to reject history items");
  Sampler* s = (Sampler*)h;
  const int G = n_neg + 1;
  std::vector<int64_t> hits;
  for (int64_t r = 0; r < n; ++r) {
    const int64_t u = user_id[r], pos = pos_item[r];
    const bool known = hist_ptr && u >= 0 && u < n_users && hist_ptr[u + 1] > hist_ptr[u];
    const int32_t* sb = (known && hist_sorted) ? hist_sorted + hist_ptr[u] : nullptr;
    const int32_t* se = (known && hist_sorted) ? hist_sorted + hist_ptr[u + 1] : nullptr;
    int64_t* out = item_id + r * G;
    out[0] = pos;
    for (int k = 1; k < G; ++k) {   // addnegsamples.py:97-108
      int64_t picked = 0;
      for (int tries = 100; tries > 0; --tries) {
        const int64_t c = draw_one(s, n_items);
        if (c != pos && !(reject_history && sb && in_sorted(sb, se, c))) { picked = c; break; }
      }
      out[k] = picked;
    }
    if (!item_seq) continue;
    // history transform (adduserhistory.py:32-73); `items` = the whole id group incl. negatives (seqrecdataset.py:45)
    const int32_t* hb = known ? hist_items + hist_ptr[u] : nullptr;
    int64_t hl = known ? hist_ptr[u + 1] - hist_ptr[u] : 1;   // unknown user: history = [0]
    int32_t* seq = item_seq + r * (int64_t)L;
    memset(seq, 0, sizeof(int32_t) * L);
    auto in_group = [&](int32_t it) { for (int k = 0; k < G; ++k) if (out[k] == it) return true; return false; };
    if (mask_mode == 1 && known) {
      hits.clear();
      for (int64_t i = 0; i < hl; ++i) if (in_group(hb[i])) hits.push_back(i);
      if (!hits.empty()) hl = seq_last ? hits.back() : hits[s->rng.randbelow(hits.size())];   // random.choice(n)
    }
    // _padding (seqrecdataset.py:60-68): last L items, left-padded
    const int64_t take = std::min<int64_t>(hl, L), start = hl - take;
    for (int64_t i = 0; i < take; ++i) {
      int32_t it = known ? hb[start + i] : 0;
      if (mask_mode == 0 && known && in_group(it)) it = 0;   // 'unorder': target ids zeroed inside the history
      seq[L - take + i] = it;
    }
    seq_len[r] = take;   // min(len(history), L), seqrecdataset.py:47
  }
  return UR_OK;
}
