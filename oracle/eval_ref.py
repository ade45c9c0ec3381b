"""CPU oracle: one_vs_all full-item ranking.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates Evaluator.evaluate_with_full_items (unirec/facility/evaluation/evaluator_abc.py:189-278) and
get_rank / the rank-derived metrics (unirec/facility/evaluation/onepos.py:20-31, 100-175) in numpy:
dense scores, history masked to NINF=-9999, column 0 <- target score, target column <- NINF, then
rank = #{columns 1.. with score > column 0}.  The reference's +-1e-8 uniform tie-breaking noise
(onepos.py:118-122, unseeded per shape) is NOT restated; pinned against tests/golden/g11_*.npz.
"""
import numpy as np

NINF = -9999.0   # evaluator_abc.py:43


def full_scores(user_emb, item_emb, user_bias=None, item_bias=None, tau=1.0, dtype=np.float32):
    """evaluator_abc.py:232-251: (U @ E^T + item_bias + user_bias) / tau."""
    s = user_emb.astype(dtype) @ item_emb.astype(dtype).T
    if item_bias is not None:
        s = s + item_bias.reshape(1, -1).astype(dtype)
    if user_bias is not None:
        s = s + user_bias.reshape(-1, 1).astype(dtype)
    return s / dtype(tau)


def full_rank(scores, user_id, item_id, user_history):
    """evaluator_abc.py:253-268 + onepos.py:20-31.  scores is modified in place, like the reference does.
    Returns (rank int32[B], target_score[B])."""
    B = scores.shape[0]
    rank = np.empty(B, dtype=np.int32)
    ts = np.empty(B, dtype=scores.dtype)
    for b in range(B):
        t = int(item_id[b])
        ts[b] = scores[b, t]
        u = int(user_id[b]) if user_id is not None else -1
        if user_history is not None and 0 <= u < len(user_history) and user_history[u] is not None:
            scores[b][np.asarray(user_history[u], dtype=np.int64)] = NINF
        scores[b, 0] = ts[b]
        scores[b, t] = NINF
        rank[b] = int((scores[b, 1:] > scores[b, 0]).sum())
    return rank, ts


def near_ties(scores64, user_id, item_id, margin):
    """Number of items per row whose fp64 score is within `margin` of the target's: the rank may legitimately differ
    by at most this many between two fp32 evaluation orders."""
    t = scores64[np.arange(len(item_id)), item_id]
    return (np.abs(scores64 - t[:, None]) < margin).sum(1) - 1


def metrics_from_rank(rank, n_scores, ks=(1, 5, 10)):
    """onepos.py:100-175 (hit@k, ndcg@k, mrr, group_auc), averaged over rows (merge_scores)."""
    r = np.asarray(rank, dtype=np.float64)
    out = {"mrr": float(np.mean(1.0 / (r + 1))), "group_auc": float(np.mean((n_scores - 1 - r) / (n_scores - 1)))}
    for k in ks:
        out[f"hit@{k}"] = float(np.mean(r < k))
        out[f"ndcg@{k}"] = float(np.mean(np.where(r < k, 1.0 / np.log2(r + 2), 0.0)))
    return out


def full_topk(scores, k, user_hist_rows=None):
    """BaseRecommender.topk with candidates=None (unirec/model/base/recommender.py:149-197): scores[row, history] = -inf,
    then the k best per row, best first (ties: smaller id first, as torch.topk does not specify).  Item 0 is masked too
    (the reference masks it through the zero padding of user_hist).  scores is modified in place.
    user_hist_rows: list of per-row id arrays (or None).  -> (scores [B,k], ids int64 [B,k])."""
    B, N = scores.shape
    scores[:, 0] = -np.inf
    if user_hist_rows is not None:
        for b, h in enumerate(user_hist_rows):
            if h is not None and len(h):
                scores[b, np.asarray(h, dtype=np.int64)] = -np.inf
    order = np.lexsort((np.broadcast_to(np.arange(N), scores.shape), -scores), axis=1)[:, :k]
    top = np.take_along_axis(scores, order, axis=1)
    ids = np.where(np.isinf(top), -1, order)
    if k > N:   # fewer items than k: pad with (-inf, -1)
        top = np.concatenate([top, np.full((B, k - N), -np.inf, dtype=top.dtype)], 1)
        ids = np.concatenate([ids, np.full((B, k - N), -1, dtype=ids.dtype)], 1)
    return top, ids.astype(np.int64)
