"""Small host utilities mirrored from unirec/utils/general.py (init_seed :26-39, get_class_instance :74-103)."""
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
