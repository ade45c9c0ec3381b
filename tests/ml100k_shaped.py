"""ML-100K-SHAPED synthetic interactions (BASELINE.json configs[0] = C1: 943 users, 1 682 items, 100 000 (user, item) rows), written in the
reference's on-disk formats (pickled DataFrames + data.info: unirec/utils/general.py:111-149).  ONE generator, seeded: tools/capture_goldens.py
feeds these files to the reference's own Trainer.fit (golden g10_trainer_fit_mf_c1), tests/test_trainer_gpu.py regenerates the SAME files
for the HIP path.  Data only -- no reference code."""
import json
import os

import numpy as np
import pandas as pd

N_USERS, N_ITEMS, N_ROWS = 944, 1683, 100_000     # ids 1..943 / 1..1682; 0 = padding


def interactions(seed=100, n_rows=N_ROWS):
    rng = np.random.default_rng(seed)
    users = 1 + (rng.random(n_rows) ** 1.5 * (N_USERS - 1)).astype(np.int64)      # a few heavy users
    items = 1 + (rng.random(n_rows) ** 2.5 * (N_ITEMS - 1)).astype(np.int64)      # a popular head
    return np.minimum(users, N_USERS - 1), np.minimum(items, N_ITEMS - 1)


def write(ddir, seed=100, n_rows=N_ROWS):
    os.makedirs(ddir, exist_ok=True)
    users, items = interactions(seed, n_rows)
    df = pd.DataFrame({"user_id": users, "item_id": items})
    df.to_pickle(os.path.join(ddir, "train.pkl"))
    df.to_pickle(os.path.join(ddir, "user_history.pkl"))
    with open(os.path.join(ddir, "data.info"), "w") as f:
        json.dump({"n_users": N_USERS, "n_items": N_ITEMS, "train_file_format": "user-item", "user_history_file_format": "user-item"}, f)
    return ddir


def initial_state(seed=101, d=64, std=0.02):
    """MF's two tables (row 0 = padding = zeros), the same numbers on both sides: nothing large to commit"""
    rng = np.random.default_rng(seed)
    u = (rng.standard_normal((N_USERS, d)) * std).astype(np.float32)
    i = (rng.standard_normal((N_ITEMS, d)) * std).astype(np.float32)
    u[0] = 0
    i[0] = 0
    return {"user_embedding.weight": u, "item_embedding.weight": i}
