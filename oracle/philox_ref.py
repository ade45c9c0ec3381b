"""numpy restatement of the DEVICE negative sampler (oracle; test infrastructure only).

Rule restated from the reference (unirec/data/transform/addnegsamples.py:67-80,97-108): uniform over [1, N-1],
reject the row's positive and the user's history, at most 100 tries, else id 0.  The random source is NOT the
reference's (CPython MT19937, reproduced by oracle/data_ref.py and unirec_amd's host sampler) but the
counter-based Philox4x32-10 (Salmon et al., SC'11; constants as in Random123), keyed by the 64-bit seed with
counter (step, row, slot, try) -- so "parity" for this kernel means bit-exact against THIS restatement plus the
distributional checks in tests/ (uniformity chi-square, no history hits).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """vectorised over equal-shaped uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3))
    k0, k1 = int(k0), int(k1)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def sample_negatives(user_id, pos_item, K, n_items, hist_ptr=None, hist_sorted=None, seed=0, step=0, alias=None):
    """-> item_id int64[B, K+1] with the positive in column 0.  alias: None or (odds, idx) -- the popularity-biased draw of
    unirec/utils/sampling.py:26-30 (x = random()*N; i = int(x); idx[i] if x - i > odds[i] else i) with random() =
    ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53 from the try's first two Philox words (CPython's construction)."""
    pos_item = np.asarray(pos_item, dtype=np.int64)
    B = len(pos_item)
    out = np.zeros((B, K + 1), dtype=np.int64)
    out[:, 0] = pos_item
    rng_range = n_items - 1
    bits = int(rng_range).bit_length()
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for b in range(B):
        hist = ()
        if hist_ptr is not None and user_id is not None and 0 <= user_id[b] < len(hist_ptr) - 1:
            hist = hist_sorted[hist_ptr[user_id[b]]:hist_ptr[user_id[b] + 1]]
        hist = set(int(x) for x in hist)
        for k in range(1, K + 1):
            tr = np.arange(100, dtype=np.uint64)
            w = philox4x32_10(np.full(100, step), np.full(100, b), np.full(100, k - 1), tr, k0, k1)
            for t in range(100):
                r = rng_range
                for q in range(4):
                    c = int(w[q][t]) >> (32 - bits)
                    if r >= rng_range and c < rng_range:
                        r = c
                if r >= rng_range:
                    r = (int(w[3][t]) * rng_range) >> 32
                cand = 1 + r
                if alias is not None:
                    u = ((int(w[0][t]) >> 5) * 67108864.0 + (int(w[1][t]) >> 6)) / 9007199254740992.0
                    x = u * n_items
                    i = int(x)
                    cand = int(alias[1][i]) if (x - i) > alias[0][i] else i
                    if cand <= 0:
                        continue
                if cand != int(pos_item[b]) and cand not in hist:
                    out[b, k] = cand
                    break
    return out


def build_seq(user_id, item_id, hist_ptr, hist_items, L, mask_mode="autoregressive", seq_last=0, match_all=False, seed=0, step=0):
    """numpy restatement of ur_device_build_seq: AddUserHistory (unirec/data/transform/adduserhistory.py:32-73) + left
    padding (unirec/data/dataset/seqrecdataset.py:60-68); the autoregressive cut picks occurrence
    (philox(step, row, 0xFFFFFFFF, 0)[0] * count) >> 32 unless seq_last.  -> (item_seq int32[B,L], seq_len int64[B])."""
    item_id = np.asarray(item_id, dtype=np.int64)
    B = len(user_id)
    n_users = len(hist_ptr) - 1
    seq = np.zeros((B, L), dtype=np.int32)
    slen = np.zeros(B, dtype=np.int64)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for b in range(B):
        u = int(user_id[b])
        hist = np.asarray(hist_items[hist_ptr[u]:hist_ptr[u + 1]], dtype=np.int32) if 0 <= u < n_users else np.zeros(0, np.int32)
        if len(hist) == 0:
            slen[b] = min(1, L)          # the reference substitutes [0]
            continue
        ids = set(int(x) for x in item_id[b]) if match_all else {int(item_id[b, 0])}
        hit = np.array([int(h) in ids for h in hist])
        if mask_mode == "unorder":
            hist = np.where(hit, 0, hist).astype(np.int32)
        elif mask_mode == "autoregressive" and hit.any():
            occ = np.flatnonzero(hit)
            if seq_last:
                t = len(occ) - 1
            else:
                w = philox4x32_10([step], [b], [0xFFFFFFFF], [0], k0, k1)
                t = (int(w[0][0]) * len(occ)) >> 32
            hist = hist[: occ[t]]
        n = len(hist)
        if n >= L:
            seq[b] = hist[n - L:]
        elif n:
            seq[b, L - n:] = hist
        slen[b] = min(n, L)
    return seq, slen
