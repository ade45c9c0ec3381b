"""SeqRecDataset -- mirror of unirec/data/dataset/seqrecdataset.py:20-68: appends item_seq / item_seq_len
(history transform + left padding to max_seq_len) to BaseDataset's rows."""
from ..rows import HistoryCSR, HostRowBuilder
from .basedataset import BaseDataset


class SeqRecDataset(BaseDataset):
    def __init__(self, config, path=None, filename=None, transform=None, data=None):
        super().__init__(config, path, filename, transform, data)
        self.add_seq_transform = None
        self._seq_builder = None

    def set_return_column_index(self):
        super().set_return_column_index()
        self.return_key_2_index["item_seq"] = len(self.return_key_2_index)
        self.return_key_2_index["item_seq_len"] = len(self.return_key_2_index)

    def add_user_history_transform(self, transform):
        """transform: AddUserHistory (its user2history / mask_mode / seq_last configure the native builder)."""
        self.add_seq_transform = transform
        neg = self.transform
        u2h = transform.user2history
        hist = u2h if isinstance(u2h, HistoryCSR) else HistoryCSR(u2h)
        b = neg.builder
        # ONE stream for negatives and history cuts, as the reference's global `random`
        self._seq_builder = HostRowBuilder(b.n_users, b.n_items, b.n_neg, self.config["max_seq_len"], hist,
                                           reject_history=neg.history is not None, mask_mode=transform.mask_mode,
                                           seq_last=transform.seq_last, seed=self.config.get("seed", 2022))
        if neg.history is not None and neg.history is not hist:
            self._seq_builder.history = hist

    def _builder(self):
        if self._seq_builder is None:
            raise RuntimeError("call add_user_history_transform() first")
        return self._seq_builder

    def __getitem__(self, index):
        u, it = self.dataset[index]
        r = self._builder().build([u], [it])
        return int(u), r["item_id"][0], r["label"][0], r["item_seq"][0], int(r["item_seq_len"][0])

    def get_batch(self, indices):
        sel = self.dataset[indices]
        return self._builder().build(sel[:, 0], sel[:, 1])
