"""Process-group indirection of the multi-GPU step + the in-process LOOPBACK group.

``facility/distributed.py``, ``sharded.py`` and ``facility/trainer.py`` talk to "the ranks" through the functions of this module, which
have torch.distributed's names and signatures.  With a torch.distributed group (or None: the default group) they ARE torch.distributed's.
With a ``LoopbackGroup`` the W ranks are W THREADS of this process sharing one GPU -- each with its own model shard, optimizer and streams:

  * the step's hot collectives (``LoopbackGroup.all_to_all`` / ``all_reduce_sum``: the fixed-capacity exchanges and the dense all-reduce)
    go through the library's loopback transport (include/unirec_amd.h: ur_loop_*): stream-ordered device-to-device copies behind
    cross-rank events, a host rendezvous of the rank threads between the three enqueue phases, NO device-host synchronisation -- the
    stream semantics an RCCL run has, on a 1-GPU box (gloo stages every block through the host and so serialises the schedule);
  * everything else (parameter broadcast at construction, checkpoints, evaluation, object gathers) is a plain rendezvous through host
    slots with full synchronisation: correctness plumbing, not the path under test.

The reference's counterpart is a real 2-process NCCL run (tests/test_model/run_ddp_test.sh:28-86); this is what a single device allows."""
import queue
import threading

import torch
import torch.distributed as _td

from . import _lib
from ._lib import check, lib

ReduceOp = _td.ReduceOp
is_available = _td.is_available


class LoopbackGroup:
    """W ranks = W threads of this process.  Every rank thread calls ``attach(rank)`` first (and ``detach()`` at the end)."""

    TIMEOUT = 180.0     # seconds a rank waits for its peers at a rendezvous: a rank that died must not hang the others for ever

    def __init__(self, world):
        self.world = int(world)
        self.handle = lib.ur_loop_create(self.world)
        if not self.handle:
            raise _lib.UnirecAmdError("ur_loop_create failed")
        self._bar = threading.Barrier(self.world, timeout=self.TIMEOUT)
        self._slots = [None] * self.world
        self._tls = threading.local()
        self._p2p = {}
        self._p2p_lock = threading.Lock()
        _lib.serialize_calls(True)      # the library's lazily initialised per-device state is not thread-safe: one call at a time

    # ---- membership
    def attach(self, rank, stream=None):
        """bind the calling thread to `rank`; it works on `stream` (a fresh one by default: every rank has its own main stream)"""
        self._tls.rank = int(rank)
        check(lib.ur_loop_attach(self.handle, int(rank)), "ur_loop_attach")
        self._tls.stream = stream if stream is not None else torch.cuda.Stream()
        torch.cuda.set_stream(self._tls.stream)
        return self

    def detach(self):
        check(lib.ur_loop_detach(), "ur_loop_detach")
        torch.cuda.set_stream(torch.cuda.default_stream())

    def abort(self):
        """a rank failed: release the peers waiting at a rendezvous (they raise BrokenBarrierError)"""
        self._bar.abort()

    @property
    def rank(self):
        return self._tls.rank

    def _stream(self):
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- the hot collectives: stream-ordered, no device-host synchronisation
    def all_to_all(self, send, recv, ahead=False, kind="rows"):
        """equal-split all-to-all of a packed block (block p of `send` -> rank p) on the CURRENT stream; ahead: communicator index 1"""
        import ctypes as C
        assert send.is_contiguous() and recv.is_contiguous() and send.numel() == recv.numel() and send.numel() % self.world == 0
        assert ahead is not None, "which communicator: ahead=True (work issued a step ahead) or False (the step's own exchanges)"
        comm = 1 if ahead else 0
        st = self._stream()
        try:
            check(lib.ur_loop_post(C.c_void_p(send.data_ptr()), comm, st), "ur_loop_post")
            self._bar.wait()
            check(lib.ur_loop_all_to_all_pull(C.c_void_p(recv.data_ptr()), send.numel() * send.element_size() // self.world,
                                              {"ids": 0, "rows": 1, "grads": 2}[kind], comm, st), "ur_loop_all_to_all_pull")
            self._bar.wait()
            check(lib.ur_loop_finish(comm, None, 0, st), "ur_loop_finish")
        except threading.BrokenBarrierError:
            raise
        except BaseException:
            self.abort()       # a rank that fails between post and finish releases its peers at once (not after TIMEOUT seconds)
            raise
        return recv

    def all_reduce_sum(self, t, ahead=True):
        """in-place sum over the ranks (rank order: identical bits everywhere) on the CURRENT stream"""
        import ctypes as C
        assert t.is_contiguous() and t.dtype == torch.float32
        comm = 1 if ahead else 0
        st = self._stream()
        try:
            check(lib.ur_loop_post(C.c_void_p(t.data_ptr()), comm, st), "ur_loop_post")
            self._bar.wait()
            check(lib.ur_loop_all_reduce_pull(t.numel(), comm, st), "ur_loop_all_reduce_pull")
            self._bar.wait()
            check(lib.ur_loop_finish(comm, C.c_void_p(t.data_ptr()), t.numel(), st), "ur_loop_finish")
        except threading.BrokenBarrierError:
            raise
        except BaseException:
            self.abort()
            raise
        return t

    # ---- everything else: a synchronised rendezvous through host slots
    def _exchange(self, obj):
        """-> [every rank's obj, in rank order]; device tensors are complete when published and must be consumed (copied) before the
        caller returns to code that may overwrite them: see the second rendezvous in the callers"""
        if torch.is_tensor(obj) and obj.is_cuda:
            torch.cuda.current_stream().synchronize()
        self._slots[self.rank] = obj
        self._bar.wait()
        out = list(self._slots)
        return out

    def _done(self):
        torch.cuda.current_stream().synchronize()
        self._bar.wait()

    def barrier(self):
        torch.cuda.current_stream().synchronize()
        self._bar.wait()

    def broadcast(self, t, src=0):
        xs = self._exchange(t if self.rank == src else None)
        if self.rank != src:
            t.copy_(xs[src])
        self._done()

    def all_reduce(self, t, op=None):
        xs = self._exchange(t.clone())
        red = {None: torch.add, _td.ReduceOp.SUM: torch.add, _td.ReduceOp.MIN: torch.minimum, _td.ReduceOp.MAX: torch.maximum}[op]
        acc = xs[0].to(t.device).clone()
        for x in xs[1:]:
            acc = red(acc, x.to(t.device))
        t.copy_(acc)
        self._done()

    def all_gather(self, parts, t):
        xs = self._exchange(t)
        for p, x in zip(parts, xs):
            p.copy_(x)
        self._done()

    def all_to_all_single(self, out, inp):
        xs = self._exchange(inp)
        W = self.world
        o = out.view(W, -1)
        for p in range(W):
            o[p].copy_(xs[p].reshape(W, -1)[self.rank])
        self._done()

    def all_gather_object(self, box, obj):
        xs = self._exchange(obj)
        for i, x in enumerate(xs):
            box[i] = x
        self._bar.wait()

    def broadcast_object_list(self, box, src=0):
        xs = self._exchange(list(box) if self.rank == src else None)
        if self.rank != src:
            for i, x in enumerate(xs[src]):
                box[i] = x
        self._bar.wait()

    def _chan(self, src, dst):
        with self._p2p_lock:
            return self._p2p.setdefault((src, dst), queue.Queue())

    def send(self, t, dst):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self._chan(self.rank, dst).put(t.detach().clone())

    def recv(self, t, src):
        x = self._chan(src, self.rank).get(timeout=self.TIMEOUT)
        t.copy_(x.to(t.device))
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()

    def close(self):
        if self.handle:
            lib.ur_loop_destroy(self.handle)
            self.handle = None
            _lib.serialize_calls(False)     # (reference-counted: the raw entry points come back with the last open group)


def is_loopback(group):
    return isinstance(group, LoopbackGroup)


# ---- torch.distributed's surface, dispatched on the group type
def is_initialized():
    return _td.is_initialized()


def get_rank(group=None):
    return group.rank if is_loopback(group) else _td.get_rank(group)


def get_world_size(group=None):
    return group.world if is_loopback(group) else _td.get_world_size(group)


def get_backend(group=None):
    return "loopback" if is_loopback(group) else _td.get_backend(group)


def barrier(group=None):
    return group.barrier() if is_loopback(group) else _td.barrier(group=group)


def broadcast(t, src=0, group=None):
    return group.broadcast(t, src) if is_loopback(group) else _td.broadcast(t, src=src, group=group)


def all_reduce(t, op=_td.ReduceOp.SUM, group=None):
    return group.all_reduce(t, op) if is_loopback(group) else _td.all_reduce(t, op=op, group=group)


def all_gather(parts, t, group=None):
    return group.all_gather(parts, t) if is_loopback(group) else _td.all_gather(parts, t, group=group)


def all_to_all_single(out, inp, group=None):
    return group.all_to_all_single(out, inp) if is_loopback(group) else _td.all_to_all_single(out, inp, group=group)


def all_gather_object(box, obj, group=None):
    return group.all_gather_object(box, obj) if is_loopback(group) else _td.all_gather_object(box, obj, group=group)


def broadcast_object_list(box, src=0, group=None):
    return group.broadcast_object_list(box, src) if is_loopback(group) else _td.broadcast_object_list(box, src=src, group=group)


def send(t, dst, group=None):
    return group.send(t, dst) if is_loopback(group) else _td.send(t, dst=dst, group=group)


def recv(t, src, group=None):
    return group.recv(t, src) if is_loopback(group) else _td.recv(t, src=src, group=group)
