"""Constants baked into the hot-path math (reference: unirec/constants/*.py)."""
EPS = 1e-8              # global_variables.py:4
VALID_TRIGGER_P = 0.1   # global_variables.py:6 (label sanity check frequency; no numerical effect)
LOSS_TYPES = ("bce", "bpr", "softmax", "ccl", "fullsoftmax")       # loss_funcs.py:6-11
HISTORY_MASK_MODES = ("unorder", "autoregressive")                 # protocols.py HistoryMaskMode
