"""Batched, native construction of training rows (host C++ and device HIP) behind the reference's transform names.

``HistoryCSR`` is the flat form of the reference's ``user2history`` (object ndarray of per-user int32 arrays,
unirec/utils/general.py:111-149): ptr[n_users+1], items (interaction order), sorted (membership tests).
"""
import ctypes as C

import numpy as np
import torch

from .. import ops
from .._lib import check, lib

MASK_MODES = {"unorder": 0, "autoregressive": 1}


class HistoryCSR:
    def __init__(self, user2history, n_users=None):
        n = len(user2history) if n_users is None else n_users
        lens = np.zeros(n, dtype=np.int64)
        for u in range(min(n, len(user2history))):
            h = user2history[u]
            lens[u] = 0 if h is None else len(h)
        self.ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=self.ptr[1:])
        self.items = np.zeros(int(self.ptr[-1]), dtype=np.int32)
        for u in range(min(n, len(user2history))):
            if lens[u]:
                self.items[self.ptr[u]:self.ptr[u + 1]] = np.asarray(user2history[u], dtype=np.int32)
        self.sorted = self.items.copy()
        for u in range(n):
            if lens[u] > 1:
                self.sorted[self.ptr[u]:self.ptr[u + 1]].sort()
        self.n_users = n
        self._dev = None

    # ---- flat binary form (convert a pickled user_history once, reload in milliseconds) ----------------------
    def save(self, filename):
        """<filename>.npz with ptr int64[n_users+1] and items int32[nnz] in per-user (time) order."""
        np.savez(filename, ptr=self.ptr, items=self.items)

    @classmethod
    def load(cls, filename):
        z = np.load(filename if str(filename).endswith(".npz") else str(filename) + ".npz")
        self = cls.__new__(cls)
        self.ptr, self.items = z["ptr"].astype(np.int64), z["items"].astype(np.int32)
        self.n_users = len(self.ptr) - 1
        self.sorted = self.items.copy()
        for u in np.flatnonzero(np.diff(self.ptr) > 1):
            self.sorted[self.ptr[u]:self.ptr[u + 1]].sort()
        self._dev = None
        return self

    def to_device(self, device):
        if self._dev is None or self._dev[0].device != torch.device(device):
            self._dev = (torch.from_numpy(self.ptr).to(device), torch.from_numpy(self.sorted).to(device))
        return self._dev

    def items_to_device(self, device):
        """hist_items (per-user time order) on the device, for the device row builder."""
        if getattr(self, "_dev_items", None) is None or self._dev_items.device != torch.device(device):
            self._dev_items = torch.from_numpy(self.items).to(device)
        return self._dev_items


def _hp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


class HostRowBuilder:
    """CPython-`random`-compatible stream + the reference's per-row rules, for a whole batch in one native call."""

    def __init__(self, n_users, n_items, n_neg, max_seq_len=0, history: HistoryCSR = None, reject_history=True,
                 mask_mode="unorder", seq_last=0, seed=2022, item_popularity=None, neg_by_pop_alpha=1.0):
        self.n_users, self.n_items, self.n_neg, self.L = n_users, n_items, n_neg, max_seq_len
        self.history, self.reject = history, bool(reject_history and history is not None)
        self.mask_mode = MASK_MODES.get(mask_mode, 2)   # unknown strings leave the history untouched (reference behaviour)
        self.seq_last = int(bool(seq_last))
        self._h = lib.ur_host_sampler_create(int(seed) & 0xFFFFFFFFFFFFFFFF)
        if item_popularity is not None:
            w = pop_sample_ratio(item_popularity, neg_by_pop_alpha)
            check(lib.ur_host_sampler_set_alias(self._h, _hp(w), len(w)), "ur_host_sampler_set_alias")

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ur_host_sampler_destroy(self._h)
            self._h = None

    def randint(self, a, b):
        return lib.ur_host_sampler_randint(self._h, a, b)

    def random(self):
        """``random.random()`` of the stream (53 bits from two 32-bit words, as CPython): what the reference's loss draws once per training
        forward for its 10 % label check (unirec/model/base/reco_abc.py:240) -- from the SAME global stream its samplers use"""
        return float(lib.ur_host_sampler_random(self._h))

    def build(self, user_id, pos_item, with_seq=True):
        """-> dict(user_id int64[n], item_id int64[n,G], label int32[n,G], item_seq int32[n,L], item_seq_len int64[n])."""
        user_id = np.ascontiguousarray(user_id, dtype=np.int64)
        pos_item = np.ascontiguousarray(pos_item, dtype=np.int64)
        n, G = len(user_id), self.n_neg + 1
        item_id = np.empty((n, G), dtype=np.int64)
        with_seq = with_seq and self.history is not None and self.L > 0
        seq = np.empty((n, self.L), dtype=np.int32) if with_seq else None
        slen = np.empty(n, dtype=np.int64) if with_seq else None
        h = self.history
        check(lib.ur_host_build_rows(self._h, _hp(user_id), _hp(pos_item), n, h.n_users if h else 0, self.n_items, self.n_neg,
                                     _hp(h.ptr) if h else None, _hp(h.items) if h else None, _hp(h.sorted) if h else None,
                                     int(self.reject), self.mask_mode, self.seq_last, self.L, _hp(item_id), _hp(seq), _hp(slen)),
              "ur_host_build_rows")
        label = np.zeros((n, G), dtype=np.int32)
        label[:, 0] = 1
        out = dict(user_id=user_id, item_id=item_id, label=label)
        if with_seq:
            out["item_seq"], out["item_seq_len"] = seq, slen
        return out


def pop_sample_ratio(item_popularity, alpha):
    """AddNegSamples._construct_item_sample_ratio (unirec/data/transform/addnegsamples.py:58-62)."""
    w = np.power(np.asarray(item_popularity).astype(float), alpha)
    w /= np.sum(w)
    w[0] = 0
    return np.ascontiguousarray(w, dtype=np.float64)


def alias_table(weights):
    """-> (odds float64[n], alias int64[n]): prepare_aliased_randomizer's table (unirec/utils/sampling.py:9-24), built natively."""
    w = np.ascontiguousarray(weights, dtype=np.float64)
    odds, alias = np.empty(len(w), dtype=np.float64), np.empty(len(w), dtype=np.int64)
    check(lib.ur_alias_table_build(_hp(w), len(w), _hp(odds), _hp(alias)), "ur_alias_table_build")
    return odds, alias


def sample_negatives_device(pos_item, K, n_items, user_id=None, history: HistoryCSR = None, seed=2022, step=0, alias=None):
    """Device sampler (Philox, order-independent): -> (item_id int64[B,K+1], label int32[B,K+1]) on pos_item's device.
    alias: None (uniform over [1, n_items-1]) or (odds float64[n_items], idx int64[n_items]) DEVICE tensors (popularity-biased)."""
    assert pos_item.is_cuda and pos_item.dtype == torch.int64
    B = pos_item.numel()
    dev = pos_item.device
    item_id = torch.empty(B, K + 1, dtype=torch.int64, device=dev)
    label = torch.empty(B, K + 1, dtype=torch.int32, device=dev)
    ptr = srt = None
    if history is not None:
        ptr, srt = history.to_device(dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
    if alias is not None:
        odds, idx = alias
        assert odds.is_cuda and odds.dtype == torch.float64 and idx.dtype == torch.int64 and odds.numel() == n_items == idx.numel()
        check(lib.ur_sample_negatives_pop(p(user_id), p(pos_item.contiguous()), B, K, n_items, history.n_users if history else 0, p(ptr),
                                          p(srt), p(odds), p(idx), int(seed) & 0xFFFFFFFFFFFFFFFF, int(step) & 0xFFFFFFFF, p(item_id),
                                          p(label), ops._stream()), "ur_sample_negatives_pop")
        return item_id, label
    check(lib.ur_sample_negatives(p(user_id), p(pos_item.contiguous()), B, K, n_items, history.n_users if history else 0, p(ptr), p(srt),
                                  int(seed) & 0xFFFFFFFFFFFFFFFF, int(step) & 0xFFFFFFFF, p(item_id), p(label),
                                  ops._stream()), "ur_sample_negatives")
    return item_id, label


class DeviceRowBuilder:
    """Device-resident input pipeline: a batch of (user, positive item) pairs already in HBM becomes
    item_id / label / item_seq / item_seq_len with two launches (ur_sample_negatives, ur_device_build_seq) and no host
    round trip.  Same rules as AddNegSamples + AddUserHistory + left padding; the random source is the counter-based
    Philox generator (order independent, keyed by (seed, step, row, slot)), not the reference's MT19937 stream -- that
    one is reproduced by HostRowBuilder."""
    MASK = {"unorder": 0, "autoregressive": 1}

    def __init__(self, n_users, n_items, n_neg, max_seq_len=0, history: HistoryCSR = None, reject_history=True,
                 mask_mode="autoregressive", seq_last=0, seed=2022, device="cuda:0", item_popularity=None, neg_by_pop_alpha=1.0,
                 rng="philox"):
        """rng="philox" (default): counter-based, order independent, 22 M rows/s.  rng="mt19937": the REFERENCE's stream -- CPython's
        `random` seeded like ``random.seed(seed)``, walked on the device row after row (ur_mt_build_rows): negatives and history cuts equal
        the reference DataLoader's (and HostRowBuilder's) bit for bit; batches must be built in the order the reference reads them."""
        self.n_users, self.n_items, self.n_neg, self.L = n_users, n_items, n_neg, max_seq_len
        if rng not in ("philox", "mt19937"):
            raise ValueError(f"rng={rng!r}: 'philox' or 'mt19937'")
        self.rng = rng
        self._mt_state, self._mt_ws = None, {}
        if rng == "mt19937" and item_popularity is not None:
            raise NotImplementedError("popularity-biased negatives draw random.random() (two words a draw): only the host builder walks that on the "
                                      "reference's stream; the device builder does with rng='philox'")
        self.alias = None
        if item_popularity is not None:       # popularity-biased negatives: the alias table lives in HBM
            odds, idx = alias_table(pop_sample_ratio(item_popularity, neg_by_pop_alpha))
            self.alias = (torch.from_numpy(odds).to(device), torch.from_numpy(idx).to(device))
        self.history, self.reject, self.seq_last, self.seed = history, bool(reject_history), int(seq_last), int(seed)
        self.mask_mode = self.MASK.get(mask_mode, 2)
        self.device = torch.device(device)
        self.step = 0

    def build(self, user_id, pos_item, with_seq=True, step=None):
        """user_id, pos_item: int64 device tensors [B]."""
        step = self.step if step is None else int(step)
        self.step = step + 1
        dev = pos_item.device
        B, G = pos_item.numel(), self.n_neg + 1
        if self.rng == "mt19937":
            return self._build_mt(user_id, pos_item, with_seq and self.L > 0)
        item_id, label = sample_negatives_device(pos_item, self.n_neg, self.n_items, user_id,
                                                 self.history if self.reject else None, self.seed, step, alias=self.alias)
        out = dict(user_id=user_id, item_id=item_id, label=label)
        if with_seq and self.L > 0:
            if self.history is None:
                raise RuntimeError("item_seq needs a user history")
            ptr, _ = self.history.to_device(dev)
            items = self.history.items_to_device(dev)
            seq = torch.empty(B, self.L, dtype=torch.int32, device=dev)
            slen = torch.empty(B, dtype=torch.int64, device=dev)
            p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
            check(lib.ur_device_build_seq(p(user_id.contiguous()), p(item_id), B, G, self.history.n_users, p(ptr), p(items), self.mask_mode,
                                          self.seq_last, 0 if self.reject else 1, self.L, self.seed & 0xFFFFFFFFFFFFFFFF,
                                          step & 0xFFFFFFFF, p(seq), p(slen), ops._stream()),
                  "ur_device_build_seq")
            out["item_seq"], out["item_seq_len"] = seq, slen
        return out

    # ---- the reference's MT19937 stream on the device
    def mt_state(self, dev):
        """uint32[626] on the device: mt[624], position, sticky error -- seeded like CPython's random.seed(seed) by the host sampler"""
        if self._mt_state is None:
            h = lib.ur_host_sampler_create(int(self.seed) & 0xFFFFFFFFFFFFFFFF)
            st = np.zeros(626, dtype=np.uint32)
            try:
                check(lib.ur_host_sampler_state(h, st.ctypes.data_as(C.c_void_p)), "ur_host_sampler_state")
            finally:
                lib.ur_host_sampler_destroy(h)
            self._mt_state = torch.from_numpy(st.view(np.int32)).to(dev)
        return self._mt_state

    def check(self):
        """raise if a batch exhausted its workspace (the kernel then left the stream where it was): synchronises"""
        if self._mt_state is not None and int(self._mt_state[625].item()) != 0:
            raise RuntimeError("DeviceRowBuilder(rng='mt19937'): a batch needed more random words than its workspace holds")

    def _build_mt(self, user_id, pos_item, with_seq):
        dev = pos_item.device
        B, K = pos_item.numel(), self.n_neg
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
        h = self.history
        want_cut = int(with_seq and h is not None and self.mask_mode == 1 and not self.seq_last)
        if with_seq and h is None:
            raise RuntimeError("item_seq needs a user history")
        key = (B, K, want_cut)
        if key not in self._mt_ws:
            nbytes = check(lib.ur_mt_workspace_bytes(B, K, self.n_items, want_cut), "ur_mt_workspace_bytes")
            self._mt_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ptr = srt = items = None
        if h is not None:
            ptr, srt = h.to_device(dev)
            items = h.items_to_device(dev)
        user_id = user_id.contiguous()
        pos_item = pos_item.contiguous()
        item_id = torch.empty(B, K + 1, dtype=torch.int64, device=dev)
        label = torch.empty(B, K + 1, dtype=torch.int32, device=dev)
        choice = torch.empty(B, dtype=torch.int32, device=dev)
        reject = int(self.reject and h is not None)
        check(lib.ur_mt_build_rows(p(self.mt_state(dev)), p(user_id), p(pos_item), B, K, self.n_items, h.n_users if h is not None else 0,
                                   p(ptr), p(items), p(srt), reject, want_cut, p(item_id), p(label), p(choice), p(self._mt_ws[key]),
                                   ops._stream()), "ur_mt_build_rows")
        out = dict(user_id=user_id, item_id=item_id, label=label)
        if with_seq:
            seq = torch.empty(B, self.L, dtype=torch.int32, device=dev)
            slen = torch.empty(B, dtype=torch.int64, device=dev)
            check(lib.ur_device_build_seq_choice(p(user_id), p(item_id), B, K + 1, h.n_users, p(ptr), p(items), self.mask_mode, self.seq_last,
                                                 0 if reject else 1, self.L, p(choice), p(seq), p(slen), ops._stream()),
                  "ur_device_build_seq_choice")
            out["item_seq"], out["item_seq_len"] = seq, slen
        return out
