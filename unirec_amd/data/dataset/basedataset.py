"""BaseDataset -- mirror of unirec/data/dataset/basedataset.py for the 'user-item' (T1) interaction format.

Differences by design: rows are produced per BATCH by the native row builder (``get_batch``) instead of per sample
in ``__getitem__`` + default_collate; ``__getitem__`` is kept (same tuple order) for drop-in use and small tests.
The on-disk format is the reference's: ``<path>/<filename>.pkl`` = pickled DataFrame with columns user_id, item_id
(examples/preprocess/prepare_data.py; loader basedataset.py:209-226)."""
import os
import pickle

import numpy as np

from ..rows import HistoryCSR, HostRowBuilder


class BaseDataset(object):
    def __init__(self, config, path=None, filename=None, transform=None, data=None):
        self.config = config
        if data is None:
            with open(os.path.join(path, filename + ".pkl"), "rb") as f:
                data = pickle.load(f)
            data = data[["user_id", "item_id"]].values
        self.dataset = np.asarray(data).astype(np.int64)
        self.transform = transform
        self.set_return_column_index()

    def set_return_column_index(self):
        self.return_key_2_index = {"user_id": 0, "item_id": 1, "label": 2}   # basedataset.py:83-87

    def __len__(self):
        return len(self.dataset)

    def _builder(self):
        t = self.transform
        if t is None:
            raise RuntimeError("training rows need an AddNegSamples transform")
        return t.builder

    def __getitem__(self, index):
        u, it = self.dataset[index]
        rows = self._builder().build([u], [it], with_seq=False)
        return int(u), rows["item_id"][0], rows["label"][0]

    def get_batch(self, indices):
        sel = self.dataset[indices]
        return self._builder().build(sel[:, 0], sel[:, 1], with_seq=False)
