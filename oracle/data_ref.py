"""Integer restatement of the reference data path (oracle; test infrastructure only).

* ``MT19937``     -- CPython's ``random.Random`` generator restated from the published MT19937
                    algorithm (Matsumoto & Nishimura 1998, init_by_array seeding as CPython's
                    Modules/_randommodule.c does) with the exact derived methods the reference
                    consumes: ``getrandbits``, ``_randbelow``, ``randint``, ``choice``, ``random``.
                    CPython is a third-party dependency that is not under /root/reference; it is
                    pinned here against the interpreter's own ``random`` module (tests) and against
                    SURVEY.md Appendix C's known answers captured from the imported reference.
* ``add_neg_samples``  -- unirec/data/transform/addnegsamples.py:90-115 (+ :67-80).
* ``add_user_history`` -- unirec/data/transform/adduserhistory.py:32-73.
* ``left_pad``         -- unirec/data/dataset/seqrecdataset.py:60-68.
* ``alias_table`` / ``alias_draw`` -- unirec/utils/sampling.py:9-31.

Pure-Python loops: only for small cases, as the oracle contract says.
"""
from typing import Iterable, List, Optional, Sequence

import numpy as np


class MT19937:
    N, M = 624, 397

    def __init__(self, seed: int = 0):
        self.mt = [0] * self.N
        self.idx = self.N
        self.seed(seed)

    # -- seeding: CPython random.seed(int) -> init_by_array(32-bit little-endian chunks of |seed|)
    def _init_genrand(self, s: int) -> None:
        mt = self.mt
        mt[0] = s & 0xFFFFFFFF
        for i in range(1, self.N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = self.N

    def seed(self, seed: int) -> None:
        a = abs(int(seed))
        key = []
        while True:
            key.append(a & 0xFFFFFFFF)
            a >>= 32
            if a == 0:
                break
        self._init_genrand(19650218)
        mt, N = self.mt, self.N
        i, j = 1, 0
        for _ in range(max(N, len(key))):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & 0xFFFFFFFF
            i += 1
            j += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
            if j >= len(key):
                j = 0
        for _ in range(N - 1):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & 0xFFFFFFFF
            i += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
        mt[0] = 0x80000000

    def _twist(self) -> None:
        mt, N, M = self.mt, self.N, self.M
        for k in range(N):
            y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7FFFFFFF)
            mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (0x9908B0DF if (y & 1) else 0)
        self.idx = 0

    def genrand_uint32(self) -> int:
        if self.idx >= self.N:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    # -- derived methods exactly as CPython's Lib/random.py / _randommodule.c
    def getrandbits(self, k: int) -> int:
        if k <= 32:
            return self.genrand_uint32() >> (32 - k)
        out, shift = 0, 0
        while k > 0:  # little-endian 32-bit words, last word truncated from the top
            r = self.genrand_uint32()
            if k < 32:
                r >>= 32 - k
            out |= r << shift
            shift += 32
            k -= 32
        return out

    def _randbelow(self, n: int) -> int:
        k = n.bit_length()
        r = self.getrandbits(k)
        while r >= n:
            r = self.getrandbits(k)
        return r

    def randint(self, a: int, b: int) -> int:
        return a + self._randbelow(b - a + 1)

    def choice(self, seq: Sequence):
        return seq[self._randbelow(len(seq))]

    def random(self) -> float:
        a = self.genrand_uint32() >> 5
        b = self.genrand_uint32() >> 6
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)


# ----------------------------------------------------------------------------- transforms
def alias_table(weights: Sequence[float]):
    """unirec/utils/sampling.py:9-24.  Returns (odds[N], alias[N]) with alias = -1 for 'None'."""
    N = len(weights)
    avg = sum(weights) / N
    odds = [1.0] * N
    alias = [-1] * N
    smalls = ((i, w / avg) for i, w in enumerate(weights) if w < avg)
    bigs = ((i, w / avg) for i, w in enumerate(weights) if w >= avg)
    small, big = next(smalls, None), next(bigs, None)
    while big and small:
        odds[small[0]], alias[small[0]] = small[1], big[0]
        big = (big[0], big[1] - (1 - small[1]))
        if big[1] < 1:
            small = big
            big = next(bigs, None)
        else:
            small = next(smalls, None)
    return odds, alias


def alias_draw(rng: MT19937, odds, alias) -> int:
    """unirec/utils/sampling.py:26-30."""
    N = len(odds)
    r = rng.random() * N
    i = int(r)
    return alias[i] if (r - i) > odds[i] else i


def pop_sample_ratio(item_popularity: np.ndarray, alpha: float) -> np.ndarray:
    """unirec/data/transform/addnegsamples.py:58-62."""
    res = np.power(item_popularity.astype(float), alpha)
    res /= np.sum(res)
    res[0] = 0
    return res


def add_neg_samples(rng: MT19937, user_id: int, pos_item, n_items: int, n_neg: int,
                    user_history_sets: Optional[Sequence[Optional[set]]] = None,
                    sampler=None) -> np.ndarray:
    """unirec/data/transform/addnegsamples.py:90-115: returns int64[pos_len + n_neg],
    positives first.  Uniform draw = random.randint(1, n_items-1) (:75-77); a candidate is
    rejected when it is a positive of this row or in the user's history (:67-73); after 100
    failed tries the slot keeps id 0 (:98-107)."""
    pos = list(pos_item) if isinstance(pos_item, (list, np.ndarray)) else [pos_item]
    pos_set = set(int(p) for p in pos)
    hist = None
    if user_history_sets is not None and user_id < len(user_history_sets):
        hist = user_history_sets[user_id]
    out = np.zeros(n_neg + len(pos), dtype=np.int64)
    for i in range(len(pos), n_neg + len(pos)):
        retries, picked = 100, 0
        while retries > 0:
            idx = rng.randint(1, n_items - 1) if sampler is None else sampler()
            if idx not in pos_set and (hist is None or idx not in hist):
                picked = idx
                break
            retries -= 1
        out[i] = picked
    out[: len(pos)] = pos
    return out


def add_user_history(rng: MT19937, user_id: int, items, user2history, mask_mode: str = "unorder",
                     seq_last: int = 0):
    """unirec/data/transform/adduserhistory.py:32-73 for data formats other than T1_1.
    Returns (history int32[len'], len')."""
    items = set(int(x) for x in items) if isinstance(items, (list, np.ndarray)) else {int(items)}
    if user_id >= len(user2history) or user2history[user_id] is None:
        history = np.zeros((1,), dtype=np.int32)
    else:
        history = np.asarray(user2history[user_id])
    if mask_mode == "unorder":
        history = history.copy()
        for i, it in enumerate(history):
            if int(it) in items:
                history[i] = 0
    elif mask_mode == "autoregressive":
        n = [i for i, it in enumerate(history) if int(it) in items]
        if n:
            cut = n[-1] if seq_last else rng.choice(n)
            history = history[:cut]
    return history, len(history)


def left_pad(x: Iterable[int], max_seq_len: int) -> np.ndarray:
    """unirec/data/dataset/seqrecdataset.py:60-68: keep the last L items, left-pad with 0."""
    x = np.asarray(list(x), dtype=np.int32)
    res = np.zeros((max_seq_len,), dtype=np.int32)
    if len(x) < max_seq_len:
        if len(x):
            res[max_seq_len - len(x):] = x
    else:
        res[:] = x[len(x) - max_seq_len:]
    return res


def make_row(rng: MT19937, user_id: int, item_id: int, *, n_items: int, n_neg: int, max_seq_len: int,
             user2history, history_sets, mask_mode: str, seq_last: int = 0):
    """SeqRecDataset.__getitem__ order of RNG consumption (basedataset.py:158-203 then
    seqrecdataset.py:38-57): negatives first, then the history cut."""
    ids = add_neg_samples(rng, user_id, item_id, n_items, n_neg, history_sets)
    label = np.zeros((n_neg + 1,), dtype=np.int32)
    label[0] = 1
    # the transform receives elements[1] = the whole id group (positive AND negatives): seqrecdataset.py:45
    hist, hl = add_user_history(rng, user_id, ids, user2history, mask_mode, seq_last)
    return user_id, ids, label, left_pad(hist, max_seq_len), min(hl, max_seq_len)
