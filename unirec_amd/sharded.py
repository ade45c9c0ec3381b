"""Routing layer of the row-sharded embedding tables: who owns which row, and torch.distributed transport of
the packed blocks when the library's own RCCL communicator is not in use.

Not in the reference: UniRec's only strategy is DDP over a replicated dense table (unirec/facility/trainer.py:67, 346) whose
[N, d] gradient is all-reduced every step.  Here row ``i`` lives on rank ``i % W`` at local row ``i // W + 1`` (local row 0 is the
padding row of every shard); the step itself is ``facility/distributed.py::ShardedSparseDenseAdam``.

``RowExchange`` is pure torch.distributed (any backend / device): the world_size-2 gloo tests drive it on the CPU.
"""
import torch

from . import pgroup as dist   # torch.distributed's surface, or the in-process loopback group's (pgroup.py)


class RowExchange:
    """Equal-split all-to-all / all-reduce / all-gather of packed blocks over a torch.distributed group (gloo has no device
    collectives: CUDA tensors are staged through the host there)."""

    def __init__(self, world: int, rank: int, group=None):
        self.world, self.rank, self.group = world, rank, group
        self.profile = None        # {label: [ms, bytes sent to peers, calls]} while profiling is on (bench.py: per-collective times)
        self._pending = []
        self.cpu_group = group if world > 1 and dist.get_backend(group) == "gloo" else None
        self.loop = group if dist.is_loopback(group) else None      # W rank threads in this process (pgroup.LoopbackGroup)

    # ---- per-collective timing (off by default: two events per collective).  bytes = what this rank sends to its W - 1 peers.
    def profile_start(self):
        self.profile, self._pending = {}, []

    def profile_stop(self):
        """-> {label: {"ms": device time of the collective on the compute stream, "MB_to_peers": ..., "calls": ...}}"""
        out = {}
        for label, e0, e1, nbytes in self._pending:
            e1.synchronize()
            rec = out.setdefault(label, {"ms": 0.0, "MB_to_peers": 0.0, "calls": 0})
            rec["ms"] += e0.elapsed_time(e1)
            rec["MB_to_peers"] += nbytes / 1e6
            rec["calls"] += 1
        self.profile, self._pending = None, []
        return out

    def _timed(self, label, nbytes, fn, on_device):
        if self.profile is None or not on_device:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self._pending.append((label, e0, e1, nbytes))
        return r

    def _staged(self, t: torch.Tensor):
        return t.is_cuda and self.world > 1 and dist.get_backend(self.group) not in ("nccl", "loopback")

    def all_to_all_equal(self, send: torch.Tensor, label="all_to_all") -> torch.Tensor:
        """send: [world * cap, ...], block p -> rank p; returns [world * cap, ...], block p <- rank p (fixed-capacity exchange)."""
        if self.world == 1:
            return send
        stage = self._staged(send)
        src = send.contiguous().cpu() if stage else send.contiguous()
        out = torch.empty_like(src)
        peers = send.numel() * send.element_size() * (self.world - 1) // self.world
        self._timed(label, peers, lambda: dist.all_to_all_single(out, src, group=self.group), send.is_cuda)
        return out.to(send.device) if stage else out

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t
        nbytes = 2 * t.numel() * t.element_size() * (self.world - 1) // self.world
        if self._staged(t):
            h = t.cpu()
            self._timed("all_reduce", nbytes, lambda: dist.all_reduce(h, group=self.group), t.is_cuda)
            return h.to(t.device)
        self._timed("all_reduce", nbytes, lambda: dist.all_reduce(t, group=self.group), t.is_cuda)
        return t

    def all_gather_cat(self, t: torch.Tensor) -> torch.Tensor:
        """equal-shaped [n, ...] from every rank -> [world * n, ...] in rank order"""
        if self.world == 1:
            return t
        src = t.contiguous().cpu() if self._staged(t) else t.contiguous()
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src, group=self.group)
        return torch.cat(parts).to(t.device)


def shard_rows(n_items: int, world: int) -> int:
    """rows per shard INCLUDING its padding row 0 (must match rows.hip build_keys_kernel)."""
    return (n_items + world - 1) // world + 1 if world > 1 else n_items


def owner_and_local(ids: torch.Tensor, world: int):
    """global id -> (owner rank, local row); id 0 (padding) -> (0, 0).  Mirrors build_keys_kernel."""
    if world == 1:
        return torch.zeros_like(ids), ids
    return torch.where(ids > 0, ids % world, torch.zeros_like(ids)), torch.where(ids > 0, ids // world + 1, torch.zeros_like(ids))


def pack_layout(counts, cap, key0_first=False):
    """Host-side statement of the fixed-capacity block layout of ur_shard_exchange_ids (tests, documentation): counts[o] unique keys
    for owner o -> [(first slot, number of keys placed)] per owner.  Slot 0 of every block is reserved; keys are right-aligned; key 0
    (the padding id: owner 0's first key when present, `key0_first`) takes no slot."""
    out = []
    for o, c in enumerate(counts):
        need = c - (1 if (o == 0 and key0_first and c > 0) else 0)
        if need > cap - 1:
            raise OverflowError(f"owner {o}: {need} keys, capacity {cap - 1}")
        out.append((o * cap + cap - need, need))
    return out
