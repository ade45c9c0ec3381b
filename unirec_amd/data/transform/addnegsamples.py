"""AddNegSamples -- same constructor and per-sample ``__call__`` as unirec/data/transform/addnegsamples.py:15,90,
backed by the native CPython-compatible stream (unirec_amd.data.rows.HostRowBuilder).  ``seed`` replaces the
reference's implicit use of the process-global ``random`` state (same sequence for the same seed)."""
import numpy as np

from ..rows import HistoryCSR, HostRowBuilder


class AddNegSamples(object):
    def __init__(self, n_users, n_items, n_neg, **kwargs):
        self.n_users, self.n_items, self.n_neg = n_users, n_items, n_neg
        u2h = kwargs.get("user2history")
        self.history = u2h if isinstance(u2h, HistoryCSR) else (HistoryCSR(u2h, max(n_users, len(u2h))) if u2h is not None else None)
        self.builder = HostRowBuilder(n_users, n_items, n_neg, 0, self.history, reject_history=True, seed=kwargs.get("seed", 2022),
                                      item_popularity=kwargs.get("item_popularity"),
                                      neg_by_pop_alpha=kwargs.get("neg_by_pop_alpha", 1.0) or 1.0)

    def __call__(self, sample):
        """sample: object ndarray [user_id, item_id, (label)] -> copy with sample[1] = int64[1 + n_neg] (positive first)."""
        sample = np.array(sample, dtype=object, copy=True)
        rows = self.builder.build([int(sample[0])], [int(sample[1])], with_seq=False)
        sample[1] = rows["item_id"][0]
        if len(sample) >= 3:
            lab = np.zeros((self.n_neg + 1,), dtype=np.int32)
            lab[0:1] = sample[2]
            sample[2] = lab
        return sample

    def sample_batch(self, user_id, item_id):
        return self.builder.build(user_id, item_id, with_seq=False)
