"""Small host utilities mirrored from unirec/utils/general.py (init_seed :26-39, get_class_instance :74-103,
load_user_history :111-149)."""
import importlib
import os
import random

import numpy as np
import torch


def init_seed(seed, reproducibility=True):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_class_instance(class_name, class_root="unirec_amd/model"):
    """Locate `<class_root>/**/<class_name.lower()>.py` and return its attribute `class_name` -- the reference's
    plug-in mechanism (models, datasets and transforms are found by NAME, nothing is registered)."""
    pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    root = os.path.join(os.path.dirname(pkg_dir), class_root) if not os.path.isabs(class_root) else class_root
    target = class_name.lower() + ".py"
    for dp, _, fns in os.walk(root):
        if target in fns:
            rel = os.path.relpath(os.path.join(dp, target[:-3]), os.path.dirname(pkg_dir))
            mod = importlib.import_module(rel.replace(os.sep, "."))
            return getattr(mod, class_name)
    raise ValueError(f"class {class_name} not found under {class_root}")


def load_user_history(file_path, file_name, n_users=None, format="user-item", time_seq=0):
    """Mirror of unirec/utils/general.py:111-149: -> (User2History, None).  User2History is an n_users-long object array;
    entry u is the ndarray of user u's items in file order (None for users without rows).
    format 'user-item' / 'user-item-rating': one interaction per row, grouped by user (rows keep their file order);
    format 'user-item_seq' / 'user-item_seq-time_seq': one row per user with the whole sequence in column item_seq."""
    import pandas as pd
    from .file_io import load_pkl_obj
    base = os.path.join(file_path, file_name)
    if os.path.exists(base + ".ftr"):
        df = pd.read_feather(base + ".ftr")
    elif os.path.exists(base + ".pkl"):
        df = load_pkl_obj(base + ".pkl")
    else:
        raise NotImplementedError(f"Unsupported user history file type: {file_name}")
    if time_seq:
        raise NotImplementedError("time sequences are not on the accelerated path")
    if n_users is None or n_users <= 0:
        n_users = int(df["user_id"].max()) + 1
    res = np.empty(n_users, dtype=object)
    if format in ("user-item", "user-item-rating"):
        users = df["user_id"].to_numpy()
        items = df["item_id"].to_numpy()
        order = np.argsort(users, kind="stable")          # groupby keeps the rows of a user in file order
        users, items = users[order], items[order]
        bounds = np.flatnonzero(np.diff(users)) + 1
        for u, seg in zip(users[np.r_[0, bounds]] if len(users) else [], np.split(items, bounds) if len(users) else []):
            res[int(u)] = seg
    elif format in ("user-item_seq", "user-item_seq-time_seq"):
        for u, seq in zip(df["user_id"].to_numpy(), df["item_seq"]):
            res[int(u)] = seq
    else:
        raise NotImplementedError(f"Unsupport user history format: {format}")
    return res, None
