"""numpy restatement of the DEVICE dropout mask generator (oracle; test infrastructure only).

The reference drops with torch's nn.Dropout (unirec/model/sequential/sasrec.py:38,69; unirec/model/modules.py:273,277,307,313,
335,352): independent Bernoulli(1-p) keeps, kept values scaled by 1/(1-p), training mode only.  WHERE the dropouts sit and
how they scale is pinned against the reference itself (tests/golden/g17_*: the reference run with recorded masks, replayed
through oracle/model_ref.py).  The random SOURCE on the device is not torch's generator but a counter-based integer hash
(csrc/common.h: DropSpec, mix32, drop_stream_key), restated here bit for bit:

    mix32(x): x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16        (uint32 arithmetic)
    key     = drop_stream_key(seed, step, site)
    keep(row, col) <=> mix32(mix32(row ^ key) + col * 0x9E3779B9) >= floor(p * 2^32)

Sites (csrc/sasrec.hip site_spec): 0 = embedded input; layer i: 4(i+1)+1 attention probabilities, +2 attention-block output,
+3 feed-forward output.  Rows are token ids b*L + l (hidden sites) or (b*H + h)*L + query position (attention), columns the
feature index / the key position.
"""
import numpy as np

U32 = np.uint32
GOLD = 0x9E3779B9


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def stream_key(seed: int, step: int, site: int) -> int:
    seed &= 0xFFFFFFFFFFFFFFFF
    step &= 0xFFFFFFFFFFFFFFFF
    k = int(mix32((seed & 0xFFFFFFFF) ^ 0x85EBCA6B))
    k = int(mix32(k ^ (seed >> 32)))
    k = int(mix32(k ^ (step & 0xFFFFFFFF)))
    k = int(mix32(k ^ (step >> 32)))
    return int(mix32((k + site * GOLD) & 0xFFFFFFFF))


def threshold(p: float) -> int:
    t = float(np.float32(p)) * 4294967296.0      # the C side multiplies the float32 probability in double
    return 4294967295 if t >= 4294967295.0 else int(t)


def mask(n_rows: int, n_cols: int, p: float, seed: int, step: int, site: int) -> np.ndarray:
    """float32 [n_rows, n_cols] multipliers: 1/(1-p) where kept, 0 where dropped (all ones when p == 0)."""
    if p <= 0:
        return np.ones((n_rows, n_cols), dtype=np.float32)
    key = stream_key(seed, step, site)
    rk = mix32(np.arange(n_rows, dtype=np.uint64) ^ np.uint64(key))[:, None]
    h = mix32((rk + np.arange(n_cols, dtype=np.uint64)[None, :] * np.uint64(GOLD)) & np.uint64(0xFFFFFFFF))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(h >= np.uint64(threshold(p)), scale, np.float32(0.0)).astype(np.float32)


def sasrec_masks(B, L, d, n_heads, n_layers, p_hidden, p_attn, seed, step):
    """the multiplier tensors of one SASRec training forward, keyed as oracle/model_ref.sasrec_user_emb(drop=...) expects."""
    m = {"embed": mask(B * L, d, p_hidden, seed, step, 0).reshape(B, L, d)}
    for i in range(n_layers):
        m[f"attn{i}"] = mask(B * n_heads * L, L, p_attn, seed, step, 4 * (i + 1) + 1).reshape(B, n_heads, L, L)
        m[f"out{i}"] = mask(B * L, d, p_hidden, seed, step, 4 * (i + 1) + 2).reshape(B, L, d)
        m[f"ffn{i}"] = mask(B * L, d, p_hidden, seed, step, 4 * (i + 1) + 3).reshape(B, L, d)
    return m


def convformer_masks(B, L, d, n_layers, p_hidden, seed, step):
    """multipliers of one ConvFormer / FASTConvFormer training forward (csrc/convformer.hip cf_site), keyed as
    oracle/model_ref.convformer_user_emb(drop=...) expects."""
    m = {"embed": mask(B * L, d, p_hidden, seed, step, 0).reshape(B, L, d)}
    for i in range(n_layers):
        m[f"out{i}"] = mask(B * L, d, p_hidden, seed, step, 4 * (i + 1) + 2).reshape(B, L, d)
        m[f"ffn{i}"] = mask(B * L, d, p_hidden, seed, step, 4 * (i + 1) + 3).reshape(B, L, d)
    return m


def gru_masks(B, L, d, p, seed, step):
    """GRU embedding dropout (csrc/gru.hip): the device rows are time-major, row id of (b, t) = t*B + b."""
    return {"embed": np.ascontiguousarray(mask(L * B, d, p, seed, step, 0).reshape(L, B, d).transpose(1, 0, 2))}


def atthist_masks(B, d, p, seed, step):
    return {"out": mask(B, d, p, seed, step, 0)}
