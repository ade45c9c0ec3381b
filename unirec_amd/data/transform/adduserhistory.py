"""AddUserHistory -- mirror of unirec/data/transform/adduserhistory.py:16,32 for the T1 ('user-item') format.
Per-sample API kept for drop-in use; training uses the batched native path (unirec_amd.data.rows)."""
import numpy as np


class AddUserHistory(object):
    def __init__(self, user2history, mask_mode="unorder", user2history_time=None, seq_last=0, data_format=None, rng=None):
        if user2history_time is not None:
            raise NotImplementedError("time sequences are outside the accelerated hot path")
        self.user2history, self.mask_mode, self.seq_last, self.data_format = user2history, mask_mode, seq_last, data_format
        self.empty_history = np.zeros((1,), dtype=np.int32)
        self.rng = rng  # a HostRowBuilder (shares the negative sampler's stream, like the reference's global `random`)

    def __call__(self, sample):
        items = sample[1]
        items = set(int(x) for x in items) if isinstance(items, (list, np.ndarray)) else {int(items)}
        u = sample[0]
        history = self.empty_history if (u >= len(self.user2history) or self.user2history[u] is None) else self.user2history[u]
        if self.mask_mode == "unorder":
            history = np.array(history, copy=True)
            for i, it in enumerate(history):
                if int(it) in items:
                    history[i] = 0
        elif self.mask_mode == "autoregressive":
            n = [i for i, it in enumerate(history) if int(it) in items]
            if n:
                if self.seq_last:
                    cut = n[-1]
                else:
                    if self.rng is None:
                        raise RuntimeError("autoregressive masking with seq_last=0 draws from the sampler stream: pass rng=")
                    cut = n[self.rng.randint(0, len(n) - 1)]   # random.choice(n) == n[_randbelow(len(n))]
                history = history[:cut]
        return history, len(history), None
