"""a dataset directory prepared by the reference's ETL (pickled DataFrames train/valid/user_history,
data.info) is read unchanged.  tests/golden/g12_dataset/ holds small files in those formats and g12_on_disk_expected.npz
what the reference's own loaders (unirec.utils.general.load_user_history, BaseDataset.load_data) returned for them
(tools/capture_goldens.py g12).  Integer data: bit-exact."""
import os

import numpy as np

from conftest import GOLDEN

DDIR = os.path.join(GOLDEN, "g12_dataset")


def _expect():
    return np.load(os.path.join(GOLDEN, "g12_on_disk_expected.npz"))


def _check_history(h, z, tag):
    ptr, items, isnone = z[tag + ".ptr"], z[tag + ".items"], z[tag + ".isnone"]
    assert len(h) == len(isnone)
    for u in range(len(h)):
        if isnone[u]:
            assert h[u] is None
        else:
            assert np.array_equal(np.asarray(h[u]).astype(np.int64), items[ptr[u]:ptr[u + 1]]), u


def test_user_history_loaders_match_reference():
    from unirec_amd.utils.file_io import load_data_info
    from unirec_amd.utils.general import load_user_history
    z = _expect()
    info = load_data_info(DDIR)
    assert info["n_users"] == int(z["n_users"]) and info["train_file_format"] == "user-item"
    h1, t1 = load_user_history(DDIR, "user_history", n_users=info["n_users"], format="user-item")
    h5, _ = load_user_history(DDIR, "user_history_seq", n_users=info["n_users"], format="user-item_seq")
    assert t1 is None
    _check_history(h1, z, "h1")
    _check_history(h5, z, "h5")
    h_inf, _ = load_user_history(DDIR, "user_history", n_users=None, format="user-item")
    assert len(h_inf) == int(z["inferred_n_users"])


def test_interaction_file_and_csr_cache(tmp_path):
    from unirec_amd.data.dataset.basedataset import BaseDataset
    from unirec_amd.data.rows import HistoryCSR
    from unirec_amd.utils.general import load_user_history
    z = _expect()
    ds = BaseDataset({}, path=DDIR, filename="train")
    assert ds.dataset.dtype == np.int64 and np.array_equal(ds.dataset, z["train"])
    h1, _ = load_user_history(DDIR, "user_history", n_users=int(z["n_users"]), format="user-item")
    csr = HistoryCSR(h1)
    assert np.array_equal(csr.ptr, z["h1.ptr"]) and np.array_equal(csr.items.astype(np.int64), z["h1.items"])
    csr.save(str(tmp_path / "hist"))
    back = HistoryCSR.load(str(tmp_path / "hist"))
    assert np.array_equal(back.ptr, csr.ptr) and np.array_equal(back.items, csr.items) and np.array_equal(back.sorted, csr.sorted)
    assert back.n_users == csr.n_users
