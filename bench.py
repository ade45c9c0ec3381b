#!/usr/bin/env python3
"""bench.py -- headline benchmark of the unirec_amd hot path (contract: see the task statement / DESIGN.md section 6).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the training hot path over one synthetic batch already resident in HBM:
ids -> sort/unique plan (+ lazy catch-up) -> SASRec encoder forward -> fused gather-dot scorer + BPR
loss -> backward (encoder, scorer) -> row-sparse gradient reduce -> Adam (dense params + touched rows).

Workload (BASELINE.json metric "SASRec seq_len=50 d=128"; configs[4]): SASRec n_items=100M, d=128,
L=50, 2 layers, 16 heads, inner 512, swish, 4 uniform negatives, BPR, per-GPU batch 512, fp32.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
HBM_COPY_GBPS = 6290.0        # MI355X_MICROARCH.md: what a streaming float4 copy reaches (79 % of the spec)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32-input MFMA peak


# algorithmic activation traffic of the GEMM classes per step at the default workload, MB
PROF_EVERY = 10   # the dominant kernel class is bracketed with HIP events on every PROF_EVERY-th timed step
# algorithmic HBM MB per step of a kernel class at the FULL row count M = B * L (scaled by the real-token fraction where used):
# row_chain = the four full-sequence chain launches: input block 3 072 B / row (item row in; x0, x0hat, q, k, v out), forward chain 6 144
# (ctx, x in; a, ahat, h1, y, yhat, next k, v out), backward chain 7 168 (gy, yhat, h1, ahat in; g_tf, g_h1, g_ta, g_ctx out), projection
# gradient 3 072 (g_qkv, g_ta, x0hat in; row gradient out) = 19 456 B / row x 25 600 rows
ALGO_BYTES_PER_STEP = {"gemm_nt": 598.0, "gemm_tn": 247.0, "row_chain": 498.0}


MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak


def split_ceiling_tflops():
    """fp32-equivalent ceiling of the split-bf16 arithmetic: the bf16 peak / the piece products per fp32 product (None: exact fp32 MFMA)"""
    from unirec_amd import ops
    t = ops.default_mfma_arith() & 0xFF
    return MFMA_BF16_PEAK_TFLOPS / t if t in (6, 9) else None


def class_terms(k):
    """piece products per fp32 product of an MFMA class's kernels (0: exact fp32-input MFMA): the weight-gradient products follow
    mfma_arith (6 / 9), the row chains run the six-term form whenever mfma_arith names a split form (test hook chain_split=0: exact)"""
    from unirec_amd import ops
    t = ops.default_mfma_arith() & 0xFF
    if t not in (6, 9):
        return 0
    if k == "gemm_tn":
        return t
    if k == "row_chain":
        return 0 if "chain_split=0" in os.environ.get("UR_TEST", "") else 6
    return 0


def class_peak_tflops(k):
    """the ceiling a class is priced against: the dense bf16 MFMA peak / its piece products when it runs split arithmetic (fp32-equivalent
    TFLOP/s), the fp32-input MFMA peak otherwise"""
    t = class_terms(k)
    return MFMA_BF16_PEAK_TFLOPS / t if t else MFMA_F32_PEAK_TFLOPS


def mfma_arith_note():
    from unirec_amd import ops
    t = ops.default_mfma_arith() & 0xFF
    if t not in (6, 9):
        return {"weight_gradients": "exact fp32-input MFMA (v_mfma_f32_32x32x2_f32)", "terms": 0}
    return {"weight_gradients": f"bf16x{t} split: fp32 operands = exact sums of three bf16 pieces, {t} piece products accumulated in fp32 on "
                                "v_mfma_f32_32x32x16_bf16; error vs fp64 <= the exact fp32-MFMA kernel's (profiles/r06_a_stage_a.txt, "
                                "tests/test_gemm_gpu.py::test_split_bf16_products_are_fp32_equivalent)",
            "terms": t, "split_ceiling_TFLOPs": round(MFMA_BF16_PEAK_TFLOPS / t, 1), "fp32_mfma_peak_TFLOPs": MFMA_F32_PEAK_TFLOPS,
            "row_chains": ("the same split, six piece products per product: weights pre-split where the K-major copies are made, the activation "
                           "fragment split in registers; error vs an fp64 evaluation of the oracle <= 1.5 x the exact chains' "
                           "(profiles/r06_k_chain_split_error.txt, tests/test_rowchain_gpu.py::test_split_row_chains_are_fp32_equivalent)")
                          if class_terms("row_chain") else "exact fp32-input MFMA",
            "everything_else": "exact fp32-input MFMA (attention, last-row layer, GRU, one-product-per-launch GEMMs)"}


def csrc_digest():
    """sha256 over the kernel sources: a committed PMC summary (tools/pmc_hbm.py writes the digest it was measured on) is only quoted
    in the line when it was taken on THIS tree's kernels"""
    import hashlib
    root = os.path.join(ROOT, "unirec_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-items", type=int, default=100_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--seq-len", type=int, default=50)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (SURVEY.md 8d)")
    ap.add_argument("--negatives", type=int, default=4)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--inner", type=int, default=512)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--loss", default="bpr")
    ap.add_argument("--table-mode", default="lazy_dense", choices=["lazy_dense", "rowwise"])
    ap.add_argument("--ids", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-bench", action="store_true")
    ap.add_argument("--autograd", action="store_true", help="step through model(...) + loss.backward() instead of forward_backward()")
    ap.add_argument("--no-prefetch", action="store_true", help="sort each batch's ids inside its own step (no side-stream lookahead)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--cpu-baseline-items", type=int, default=1_000_000)
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the steady_state and e2e legs that follow the timed region")
    ap.add_argument("--all-configs", action="store_true", help="also measure BASELINE configs C2, C3 and C4's encoder (step time, "
                    "dominant kernel class and its roofline fraction each) into the same JSON line")
    ap.add_argument("--sharded-w1", action="store_true", help="--gpus 1 only: step through the multi-GPU code path (facility/distributed.py, "
                    "fixed-capacity row exchange with world = 1, collectives degenerate) instead of the plain optimizer: its overhead")
    ap.add_argument("--loopback", type=int, default=0, help="--gpus 1 only: W rank THREADS of this process share the GPU and train ONE model "
                    "through the in-process loopback transport (unirec_amd/pgroup.py): the multi-GPU step's real schedule at true W-rank shapes "
                    "(merge plan, capacities) with real stream concurrency; prints wall time per step and per rank-step")
    ap.add_argument("--worker", action="store_true", help="(internal) world > 1: this process IS the benchmark; without it the process the "
                    "launcher started supervises a --worker child per rung of the fallback ladder (tools/bench_ladder.py)")
    ap.add_argument("--supervised", action="store_true", help="--gpus 1 only: run the ONE-rank benchmark the way a multi-GPU launch runs -- a supervisor, "
                    "a --worker child per ladder rung, the row-sharded step (--sharded-w1) through the library's RCCL communicators at world 1")
    ap.add_argument("--dry-worker", action="store_true", help="(test aid) the worker walks the phases over gloo with no GPU work")
    ap.add_argument("--no-selfcheck", action="store_true", help="world > 1: skip the W-rank == 1-rank check that runs before the timed region")
    ap.add_argument("--selfcheck-items", type=int, default=1_000_000)
    ap.add_argument("--skip-padding", type=int, default=1, help="0: the encoder carries the padded [B * L] rows (per-sequence layout) instead of "
                    "the compact real-token rows: a probe for profiles/, not a configuration of the line")
    ap.add_argument("--dropout", type=float, default=0.0, help="hidden_dropout_prob = attn_dropout_prob (the reference's SASRec.yaml "
                    "default is 0.5; its example / benchmark scripts and the headline line use 0)")
    return ap.parse_args()


def model_config(a, device):
    return dict(model="SASRec", n_users=162_542, n_items=a.n_items, device=device, loss_type=a.loss, embedding_size=a.d,
                hidden_size=a.d, dropout_prob=0.0, init_method="normal", init_mean=0.0, init_std=0.02, has_user_emb=False,
                has_user_bias=False, has_item_bias=False, distance_type="dot", tau=1.0, train_file_format="user-item",
                exp_name="bench", n_layers=a.layers, n_heads=a.heads, inner_size=a.inner, hidden_dropout_prob=a.dropout,
                attn_dropout_prob=a.dropout, seed=2022, hidden_act="swish", layer_norm_eps=1e-10, max_seq_len=a.seq_len, use_position_emb=True,
                skip_padding=getattr(a, "skip_padding", 1))


def synth_batches(a, n_items, device, seed, n_batches=8):
    """ML-25M-shaped synthetic rows generated ON DEVICE: history lengths ~ clipped log-normal
    (median ~70, so >= 50% of rows are full at L=50 and the rest are left-padded), item ids uniform over
    [1, N-1] (worst case for cache/TLB) or Zipf(1.0); K uniform negatives per row."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = []
    B, L, G = a.batch, a.seq_len, a.negatives + 1

    def draw(shape):
        if a.ids == "uniform":
            return torch.randint(1, n_items, shape, generator=g, device=device)
        u = torch.rand(shape, generator=g, device=device, dtype=torch.float64)
        return (torch.exp(u * torch.log(torch.tensor(float(n_items - 1), dtype=torch.float64, device=device)))).long().clamp_(1, n_items - 1)

    for _ in range(n_batches):
        lens = torch.exp(torch.randn(B, generator=g, device=device) * 1.0 + 4.25).clamp_(5, 1000).long().clamp_(max=L)
        seq = draw((B, L)).to(torch.int32)
        pos = torch.arange(L, device=device).unsqueeze(0)
        seq = torch.where(pos >= (L - lens).unsqueeze(1), seq, torch.zeros_like(seq)).contiguous()
        item_id = draw((B, G)).contiguous()
        label = torch.zeros(B, G, dtype=torch.int32, device=device)
        label[:, 0] = 1
        out.append(dict(item_seq=seq, item_id=item_id, label=label,
                        user_id=torch.randint(1, 162_542, (B,), generator=g, device=device)))
    return out


def cpu_baseline(a):
    """Reference semantics (dense nn.Embedding gradient, unfused attention, dense Adam over the whole table)
    restated in oracle/model_ref.py, timed on this box's host cores.  Bounded sample: same batch shape, table
    reduced to --cpu-baseline-items rows because dense Adam over 100M x 128 cannot be stepped on the host."""
    from oracle import model_ref
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    N = min(a.n_items, a.cpu_baseline_items)
    cfg = model_config(a, "cpu")
    cfg["n_items"] = N
    d, I, L = a.d, a.inner, a.seq_len
    g = torch.Generator().manual_seed(2022)
    P = {"item_embedding.weight": torch.randn(N, d, generator=g) * 0.02,
         "position_embedding.weight": torch.randn(L + 1, d, generator=g) * 0.02,
         "LayerNorm.weight": torch.ones(d), "LayerNorm.bias": torch.zeros(d)}
    for i in range(a.layers):
        pre = f"trm_encoder.layer.{i}."
        for nm in ("query", "key", "value", "dense"):
            P[pre + f"multi_head_attention.{nm}.weight"] = torch.randn(d, d, generator=g) * 0.02
            P[pre + f"multi_head_attention.{nm}.bias"] = torch.zeros(d)
        P[pre + "multi_head_attention.LayerNorm.weight"] = torch.ones(d)
        P[pre + "multi_head_attention.LayerNorm.bias"] = torch.zeros(d)
        P[pre + "feed_forward.dense_1.weight"] = torch.randn(I, d, generator=g) * 0.02
        P[pre + "feed_forward.dense_1.bias"] = torch.zeros(I)
        P[pre + "feed_forward.dense_2.weight"] = torch.randn(d, I, generator=g) * 0.02
        P[pre + "feed_forward.dense_2.bias"] = torch.zeros(d)
        P[pre + "feed_forward.LayerNorm.weight"] = torch.ones(d)
        P[pre + "feed_forward.LayerNorm.bias"] = torch.zeros(d)
    P["item_embedding.weight"][0].zero_()
    B = a.batch
    # the protocol of BASELINE.md 3 / SURVEY.md 8(d): 5 warm-up steps, then >= 20 timed steps, one DISTINCT batch per step drawn by the
    # generator the GPU run's batches come from (synth_batches, same seed; ids over the reduced table), median + p10 / p90 of the step times
    n_warm, n_timed = 5, 20
    batches = [dict(b, user_id=torch.ones(B, dtype=torch.int64)) for b in synth_batches(a, N, torch.device("cpu"), 2022, n_batches=n_warm + n_timed + 8)]
    state = {}
    # pick the thread count torch-CPU actually runs fastest with on this host (all cores is often NOT it): one probe step each
    best, probe = None, 0
    for th in sorted({min(avail, 16), min(avail, 32), min(avail, 64), avail}):
        torch.set_num_threads(th)
        model_ref.train_step(P, state, batches[probe], cfg)      # (first step at this thread count: pool start-up)
        t0 = time.perf_counter()
        model_ref.train_step(P, state, batches[probe + 1], cfg)
        one = time.perf_counter() - t0
        probe += 2
        if best is None or one < best[1]:
            best = (th, one)
        elif one > 1.2 * best[1]:
            break  # past the sweet spot: more threads only oversubscribe
    cores = best[0]
    torch.set_num_threads(cores)
    rest = batches[probe:]
    for b in rest[:n_warm]:
        model_ref.train_step(P, state, b, cfg)
    times = []
    for b in rest[n_warm:n_warm + n_timed]:
        t0 = time.perf_counter()
        model_ref.train_step(P, state, b, cfg)
        times.append(time.perf_counter() - t0)
    ts = sorted(times)
    med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    p10, p90 = ts[int(0.1 * (len(ts) - 1))], ts[int(round(0.9 * (len(ts) - 1)))]
    return {"value": round(B / med, 1), "unit": "examples/s", "cores": cores, "kind": "port",
            "protocol": {"warmup_steps": n_warm, "timed_steps": len(ts), "statistic": "median of the per-step wall times", "distinct_batches": True,
                         "ms_per_step_median": round(med * 1e3, 1), "ms_per_step_p10": round(p10 * 1e3, 1), "ms_per_step_p90": round(p90 * 1e3, 1),
                         "examples_per_s_p10_p90": [round(B / p90, 1), round(B / p10, 1)]},
            "sample": f"{len(ts)} timed steps (after {n_warm} warm-ups, one distinct batch each) of oracle/model_ref.train_step (reference semantics: "
                      f"dense [N,d] embedding gradient + dense Adam), B={B}, L={L}, d={d}, K={a.negatives}, table reduced to N={N} rows (dense Adam "
                      f"is O(N): {med * 1e3:.0f} ms/step here, would be ~{med * 1e3 * a.n_items / N:.0f} ms/step at N={a.n_items} by per-row "
                      f"extrapolation); {cores} torch threads (fastest of the counts probed, {avail} cores available); torch {torch.__version__} CPU"}


def loopback_leg(a, device, W):
    """W rank threads on ONE GPU (pgroup.LoopbackGroup): every rank holds its shard of the 100 M-row table and its optimizer state
    (W x 19 GB at W = 8), steps through ShardedSparseDenseAdam.train_step with the lookahead batch, and exchanges through stream-ordered
    device copies.  All W ranks' kernels share the one device, so `wall / W` is the device time one rank-step costs at true W-rank shapes
    (W-run merge plans, cap = 1.25 x lookups / W) including the device-to-device copies that stand in for xGMI; the copies are bracketed
    by the library's per-collective timers and reported apart."""
    import threading
    import traceback
    import ctypes as C
    from unirec_amd import _lib
    from unirec_amd.facility.distributed import ShardedSparseDenseAdam
    from unirec_amd.model.sequential.sasrec import SASRec
    from unirec_amd.pgroup import LoopbackGroup
    from unirec_amd.sharded import shard_rows
    group = LoopbackGroup(W)
    lock = threading.Lock()
    res, errs = [None] * W, [None] * W
    ncls = _lib.lib.ur_prof_num_classes()
    names = [_lib.lib.ur_prof_class_name(i).decode() for i in range(ncls)]
    coll = ("a2a_ids", "a2a_rows", "a2a_row_grads", "allreduce")

    def body(r):
        try:
            torch.cuda.set_device(device)
            group.attach(r)
            batches = synth_batches(a, a.n_items, device, 2022 + 7919 * r, n_batches=a.warmup + a.steps + 1)
            with lock:
                torch.manual_seed(2022 + r)
                model = SASRec(dict(model_config(a, str(device)), n_items=shard_rows(a.n_items, W)))
                torch.cuda.synchronize()
            opt = ShardedSparseDenseAdam(model, r, W, group=group, lr=1e-3, table_mode=a.table_mode, full_rows={"item_embedding": a.n_items})
            model.train()
            loss = None
            for i in range(a.warmup):
                loss = opt.train_step(batches[i], batches[i + 1])
            group.barrier()
            if r == 0:
                _lib.lib.ur_prof_reset()
                _lib.lib.ur_prof_set_mask(sum(1 << names.index(n) for n in coll))
                _lib.lib.ur_prof_enable(1)
            group.barrier()
            t0 = time.perf_counter()
            for i in range(a.steps):
                loss = opt.train_step(batches[a.warmup + i], batches[a.warmup + i + 1])
            torch.cuda.current_stream().synchronize()
            model.join_side_updates()
            group.barrier()
            dt = time.perf_counter() - t0
            bf = next(iter(opt._bufs.values()))
            res[r] = dict(dt=dt, loss=float(loss), cap=bf.get("cap"), cap2=bf.get("cap2"), n_overflow=opt.n_overflow)
            if r == 0:
                _lib.lib.ur_prof_enable(0)
                ms, cnt, work = (C.c_double * ncls)(), (C.c_int64 * ncls)(), (C.c_double * ncls)()
                _lib.check(_lib.lib.ur_prof_read(ms, cnt, work), "ur_prof_read")
                res[r]["copies"] = {n: dict(ms=ms[names.index(n)], calls=int(cnt[names.index(n)])) for n in coll}
                _lib.lib.ur_prof_set_mask(0xFFFFFFFF)
            opt.flush()
            torch.cuda.synchronize()
        except BaseException:   # noqa: BLE001
            errs[r] = traceback.format_exc()
            group.abort()
        finally:
            group.detach()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    bad = next((e for e in errs if e and "BrokenBarrierError" not in e), None) or next((e for e in errs if e), None)
    if bad:
        raise SystemExit("loopback leg failed:\n" + bad)
    dt = max(x["dt"] for x in res)
    copies_ms = sum(v["ms"] for v in res[0]["copies"].values())       # all ranks' brackets (one profiler per process)
    wall = dt / a.steps * 1e3
    return {"ranks": W, "steps": a.steps, "ms_per_step_wall_all_ranks_on_one_gpu": round(wall, 4), "ms_per_rank_step": round(wall / W, 4),
            "copies_ms_per_rank_step": round(copies_ms / a.steps / W, 4), "ms_per_rank_step_without_copies": round(wall / W - copies_ms / a.steps / W, 4),
            "cap": res[0]["cap"], "cap2": res[0]["cap2"], "overflows": res[0]["n_overflow"], "final_loss_rank0": round(res[0]["loss"], 6),
            "copies": {k: {"ms_per_rank_step": round(v["ms"] / a.steps / W, 4), "calls_per_rank_step": v["calls"] / a.steps / W} for k, v in res[0]["copies"].items()},
            "note": "W rank threads of ONE process on ONE GPU through the in-process loopback transport (stream-ordered device copies stand in for "
                    "xGMI); every rank's kernels share the device, so wall / W is one rank-step's device time at true W-rank shapes"}


def multi_gpu_selfcheck(a, device, rank, world):
    """world > 1, before anything is timed: 3 training steps of the SAME Trainer-level optimizer the benchmark uses
    (facility/distributed.py: row-sharded table, 3 all-to-alls, one flat all-reduce -- over whatever backend the process group
    runs, RCCL on the GPUs) must reproduce 1 rank stepping the concatenated batch: per-step loss (mean of the rank losses) and
    every parameter.  A smaller catalogue (--selfcheck-items) keeps the 1-rank reference affordable; the code path is the same.
    Raises on mismatch; returns a short report for the JSON line."""
    import numpy as np
    import torch.distributed as dist
    from unirec_amd.facility.distributed import ShardedSparseDenseAdam
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.model.sequential.sasrec import SASRec
    N = min(a.n_items, a.selfcheck_items)
    cfg = model_config(a, str(device))
    cfg["n_items"] = N
    B, Wd = a.batch, world
    torch.manual_seed(4242)
    model = SASRec(cfg)
    P0 = {k: v.detach().clone() for k, v in model.state_dict().items()} if rank == 0 else None
    opt = ShardedSparseDenseAdam(model, rank, world, lr=1e-3, table_mode=a.table_mode)
    model.train()
    full = synth_batches(argparse.Namespace(**{**vars(a), "batch": B * Wd}), N, device, 99, n_batches=3)    # same seed: same batches on every rank
    losses = []
    for i, b in enumerate(full):
        mine = {k: v[rank * B:(rank + 1) * B].contiguous() for k, v in b.items()}
        nxt = {k: v[rank * B:(rank + 1) * B].contiguous() for k, v in full[i + 1].items()} if i + 1 < len(full) else None
        losses.append(float(opt.train_step(mine, nxt)))
    all_losses = [None] * world
    dist.all_gather_object(all_losses, losses)
    # "did RCCL see N ranks" from RCCL itself: ncclCommCount / ncclCommUserRank of the library's two communicators (0 = the packed blocks
    # travel through torch.distributed on this rung); every rank must report the same count, and it must be WORLD_SIZE
    from unirec_amd import ops as _ops
    counts = [None] * world
    dist.all_gather_object(counts, _ops.comm_count())
    if any(c not in (0, world) for c in counts) or len(set(counts)) != 1:
        raise RuntimeError(f"multi-GPU self-check: RCCL reports {counts} ranks per process, torch.distributed world is {world}")
    sd = opt.gather_state_dict()
    report = None
    if rank == 0:
        m1 = SASRec(cfg)
        m1.load_state_dict(P0)
        m1.check_views()
        o1 = SparseDenseAdam(m1, lr=1e-3, table_mode=a.table_mode)
        m1.train()
        ref = []
        for b in full:
            o1.zero_grad()
            o1.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
            ref.append(float(m1.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])))
            o1.step()
        o1.flush()
        np.testing.assert_allclose(np.mean(all_losses, axis=0), ref, rtol=2e-5, err_msg="multi-GPU self-check: losses")
        worst = 0.0
        for k, v in m1.state_dict().items():
            if k.endswith("key.bias"):
                continue
            got, want = sd[k].numpy(), v.detach().cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5, err_msg=f"multi-GPU self-check: {k}")
            worst = max(worst, float(np.abs(got - want).max()))
        report = {"ok": True, "steps": 3, "n_items": N, "ranks": world, "rccl_ranks": counts[0], "backend": dist.get_backend(),
                  "max_abs_param_diff_vs_1_rank": worst, "losses": [round(float(x), 6) for x in np.mean(all_losses, axis=0)]}
        del m1, o1
    del model, opt, sd
    torch.cuda.empty_cache()
    dist.barrier()
    return report


MFMA_CLASSES = ("gemm_nt", "gemm_tn", "row_chain", "row_chain_last", "attn_fwd", "attn_bwd", "gru")   # classes whose `work` is flops; the rest count bytes


def steady_state_leg(a, opt, step_fn, batches, steps, barrier):
    """The headline steps meet rows that were never updated (N = 100 M, ~27 K rows per step), so the lazily-evaluated dense Adam
    has nothing to replay.  In a long run every looked-up row HAS history -- a row comes back every ~N / 27 K steps -- and the
    replay of its missed zero-gradient steps is part of every step.  This leg ages the whole table (every row: moments set,
    last update `gap` steps ago) and times the same step again; `embedding_optimizer: rowwise` has no replay at all."""
    st = opt.tables["item_embedding"]
    if st["last"] is None:
        return {"note": "rowwise table optimizer: no replay, steady state == headline"}
    gap = max(200, min(4000, a.n_items // 27_000))
    st["m"].fill_(1e-5)
    st["v"].fill_(1e-10)
    st["m"][0].zero_()
    st["v"][0].zero_()
    opt.t += gap
    st["last"].fill_(opt.t - gap)
    for i in range(3):
        step_fn(batches[i % len(batches)], batches[(i + 1) % len(batches)])
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(batches[(3 + i) % len(batches)], batches[(4 + i) % len(batches)])
    barrier()
    dt = time.perf_counter() - t0
    return {"ms_per_step": round(dt / steps * 1e3, 4), "examples_per_s": round(a.batch * steps / dt, 1), "steps": steps,
            "replayed_gap_steps": gap, "note": "every looked-up row replays the zero-gradient Adam steps it missed (capped at 192 terms)"}


def e2e_leg(a, model, opt, device, steps):
    """The input pipeline INSIDE the timed loop (north_star: the uniform negative sampler is part of the path): interaction pairs
    and the CSR history live in HBM; per step ur_sample_negatives + ur_device_build_seq (DeviceRowBuilder: negatives rejected
    against the user's history, history cut at the target, left padding) build the batch, then the same training step runs.
    Comparable with the headline: the target of every pair is the user's LAST interaction and the histories are one item longer than
    synth_batches' lengths, so the cut histories have the headline's length distribution (`real_token_fraction` says what came out)."""
    import numpy as np
    from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR
    n_users = 100_000
    rng = np.random.default_rng(0)
    lens = np.clip(np.exp(rng.normal(4.25, 1.0, n_users)).astype(np.int64), 5, 1000) + 1
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    items = rng.integers(1, a.n_items, int(ptr[-1])).astype(np.int32)
    csr = HistoryCSR.__new__(HistoryCSR)
    csr.ptr, csr.items, csr.n_users, csr._dev = ptr, items, n_users, None
    order = np.lexsort((items, np.repeat(np.arange(n_users), lens)))
    csr.sorted = items[order]
    n_pairs = a.batch * (steps + 6)
    users = rng.integers(0, n_users, n_pairs)
    pos = items[ptr[users] + lens[users] - 1]
    pairs = torch.from_numpy(np.stack([users, pos.astype(np.int64)], 1)).to(device)
    bld = DeviceRowBuilder(n_users, a.n_items, a.negatives, a.seq_len, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=1,
                           device=str(device))

    # the product loader (facility/trainer.py): builds run on its own stream, two batches ahead of the step that consumes them
    from unirec_amd.facility.trainer import DeviceBatchLoader
    loader = DeviceBatchLoader(pairs, bld, a.batch, shuffle=False)
    if hasattr(opt, "plan_stream"):
        loader.use_stream(opt.plan_stream(), joined=True)     # (as Trainer.fit does: one stream for everything that runs a batch ahead)
    it = iter(loader)

    def step(b, nxt):
        opt.zero_grad()
        opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
        opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
        model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
        opt.step(late_join=True)

    cur = next(it)
    for k in range(5):
        nxt = next(it)
        step(cur, nxt)
        cur = nxt
    torch.cuda.synchronize()
    real = torch.zeros(1, device=device)
    t0 = time.perf_counter()
    for k in range(5, 5 + steps):
        nxt = next(it)              # handed out behind an event recorded a step ago; the loader queues the build of batch k + 3
        step(cur, nxt)
        cur = nxt
    t_host = time.perf_counter() - t0   # the host has queued everything: if this is close to dt, the leg is bound by the host's launch rate
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del it
    for k in range(5, 8):           # (outside the clock) what fraction of the B * L token slots these batches fill
        real += (loader._build(torch.arange(len(pairs), device=device), k, 0)["item_seq"] > 0).float().mean() / 3
    return {"ms_per_step": round(dt / steps * 1e3, 4), "examples_per_s": round(a.batch * steps / dt, 1), "steps": steps,
            "host_enqueue_ms_per_step": round(t_host / steps * 1e3, 4), "real_token_fraction": round(float(real), 4), "negatives": a.negatives,
            "pipeline": "device-resident: ur_sample_negatives (uniform, history-rejecting) + ur_device_build_seq per step, inside the clock "
                        "(DeviceBatchLoader: built on the optimizer's plan stream two batches ahead); a recurring population of 100 K users: returning rows pay the lazy replay; history lengths as the headline's"}


def trainer_fit_leg(a, device, steps=200):
    """The drop-in surface itself: ``Trainer(config, model).fit(loader)`` (unirec/facility/trainer.py:255-357) on a fresh model of the
    headline shape, fed by the DeviceBatchLoader -- fit()'s own loop (model.train() per step, the loss list, drain(), the epoch's
    bookkeeping), one epoch of `steps` batches after a short warm-up epoch, wall clock around the whole call."""
    import logging
    import numpy as np
    from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR
    from unirec_amd.facility.trainer import DeviceBatchLoader, Trainer
    from unirec_amd.model.sequential.sasrec import SASRec
    # every user is drawn ONCE (the headline's statistics: its synthetic batches are uniform ids over the 100 M rows, a row practically never
    # comes back inside a run, so the lazy-Adam replay list stays empty).  `e2e` draws users with replacement from a 100 k population
    # instead: there a history returns every epoch and the replay of ~50 rows per returning user is part of the step (+35 us).
    warm_batches = 12
    n_users = a.batch * (steps + warm_batches) if not os.environ.get("UR_FIT_RECURRING_USERS") else 100_000
    rng = np.random.default_rng(1)
    # (only the last L items before the target are ever read: histories are stored up to L + 10 long, same sequences as the e2e leg's 1000)
    lens = np.clip(np.exp(rng.normal(4.25, 1.0, n_users)).astype(np.int64), 5, a.seq_len + 10) + 1
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    items = rng.integers(1, a.n_items, int(ptr[-1])).astype(np.int32)
    csr = HistoryCSR.__new__(HistoryCSR)
    csr.ptr, csr.items, csr.n_users, csr._dev = ptr, items, n_users, None
    key = np.repeat(np.arange(n_users, dtype=np.int64), lens) << 32 | items       # per-user sorted item lists: one key sort
    key.sort()
    csr.sorted = (key & 0xFFFFFFFF).astype(np.int32)
    del key

    def loader(n_batches, seed, first=0):
        if n_users == 100_000:
            users = np.random.default_rng(seed).integers(0, n_users, a.batch * n_batches)
        else:
            users = first + np.random.default_rng(seed).permutation(a.batch * n_batches)
        pos = items[ptr[users] + lens[users] - 1]
        pairs = torch.from_numpy(np.stack([users, pos.astype(np.int64)], 1)).to(device)
        bld = DeviceRowBuilder(n_users, a.n_items, a.negatives, a.seq_len, csr, reject_history=True, mask_mode="autoregressive", seq_last=0,
                               seed=seed, device=str(device))
        return DeviceBatchLoader(pairs, bld, a.batch, shuffle=False)

    cfg = dict(model_config(a, str(device)), learning_rate=1e-3, epochs=1, optimizer="adam", scheduler="off", early_stop=0,
               embedding_optimizer=a.table_mode, output_path="/tmp/unirec_amd_bench")
    logging.getLogger(cfg.get("exp_name", "bench")).setLevel(logging.WARNING)
    torch.manual_seed(7)
    model = SASRec(cfg)
    tr = Trainer(cfg, model)
    tr.fit(loader(warm_batches, 3), valid_data=None, save_model=False)      # warm-up epoch (code objects, workspaces, the collector's freeze)
    torch.cuda.synchronize()
    data = loader(steps, 4, first=a.batch * warm_batches)
    t0 = time.perf_counter()
    tr.fit(data, valid_data=None, save_model=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the lazy rows' flush (what an evaluation or a checkpoint behind the epoch pays ONCE: a pass over the whole 100 M-row table and its
    # two moment arrays, ~300 GB of traffic) is timed apart: amortised over a 200-step epoch it would be 40 % of the "step"
    t1 = time.perf_counter()
    tr.optimizer.flush() if hasattr(tr.optimizer, "flush") else None
    torch.cuda.synchronize()
    dt_flush = time.perf_counter() - t1
    losses = tr.step_losses[-steps:]
    out = {"ms_per_step": round(dt / steps * 1e3, 4), "examples_per_s": round(a.batch * steps / dt, 1), "steps": steps,
           "final_loss": round(float(losses[-1]), 6) if losses else None, "flush_ms_once": round(dt_flush * 1e3, 2),
           # which user population the leg draws: "unique" = every user once, like the headline's uniform ids -- no returning
           # rows, so no lazy-Adam replay inside the step; "recurring-100k" (UR_FIT_RECURRING_USERS=1, and what `e2e` always does) = users with
           # replacement from 100 K, histories up to L + 10 items: the replay of returning rows is part of the step (~ + 35 us)
           "users": "recurring-100k" if n_users == 100_000 else "unique", "history_items_kept": a.seq_len + 10,
           "what": "Trainer(config, model).fit(DeviceBatchLoader) -- one epoch, wall clock around fit() (its loss list and drain() included); "
                   "flush_ms_once = optimizer.flush() behind it (every row's pending zero-gradient steps: once per evaluation / checkpoint)"}
    del tr, model
    torch.cuda.empty_cache()
    return out


def variant_leg(a, model, opt, step_fn, device, steps, dropout=None, ids=None):
    """The headline step under one of the reference's other settings, same model / table / optimizer, outside `value`:
    dropout = 0.5 at every site is what unirec/config/model/SASRec.yaml:4-5 trains with (the headline follows the reference's benchmark
    scripts: 0); ids = "zipf": item popularity ~ Zipf(1.0) instead of uniform (long runs of equal ids in the row-gradient reduce)."""
    import copy
    saved = (model.hidden_dropout_prob, model.attn_dropout_prob)
    b_args = copy.copy(a)
    if ids is not None:
        b_args.ids = ids
    batches = synth_batches(b_args, a.n_items, device, 4242, n_batches=steps + 6)
    try:
        if dropout is not None:
            model.hidden_dropout_prob = model.attn_dropout_prob = float(dropout)
            model.__dict__.pop("_ws_slots", None)      # (the dropout layout of the activation workspace)
        for i in range(4):
            step_fn(batches[i], batches[i + 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4, 4 + steps):
            step_fn(batches[i], batches[i + 1])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        model.join_side_updates()
        model.hidden_dropout_prob, model.attn_dropout_prob = saved
        if dropout is not None:
            model.__dict__.pop("_ws_slots", None)
    return {"ms_per_step": round(dt / steps * 1e3, 4), "examples_per_s": round(a.batch * steps / dt, 1), "steps": steps}


def c3_e2e_leg(device, steps=20):
    """BASELINE config C3 with its input pipeline inside the clock: the K = 1000 uniform sampler (the reference's second wall,
    unirec/data/transform/addnegsamples.py:90-115) + the history cut on the device, then the training step."""
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.model.sequential.sasrec import SASRec
    old = sys.argv
    sys.argv = [old[0], "--n-items", "2000000", "--seq-len", "200", "--negatives", "1000", "--loss", "softmax", "--batch", "128"]
    try:
        a = parse()
    finally:
        sys.argv = old
    torch.manual_seed(2022)
    model = SASRec(model_config(a, str(device)))
    opt = SparseDenseAdam(model, lr=1e-3, table_mode="lazy_dense")
    model.train()
    try:
        return e2e_leg(a, model, opt, device, steps)
    finally:
        model.join_side_updates()
        del model, opt
        torch.cuda.empty_cache()


def other_config(name, device):
    """One BASELINE configuration other than the headline one: step time + the dominant kernel class with its roofline fraction
    (same method as the headline: HIP events around every launch of the class, algorithmic work / device time)."""
    import ctypes as C
    from unirec_amd import _lib
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.model.sequential.gru import GRU
    from unirec_amd.model.sequential.sasrec import SASRec
    argv = {"C2": ["--n-items", "60000", "--d", "64"],
            "C3": ["--n-items", "2000000", "--seq-len", "200", "--negatives", "1000", "--loss", "softmax", "--batch", "128"],
            "C4_encoder": ["--n-items", "10000000", "--loss", "softmax"],
            # the reference's GRU.yaml default width (unirec/config/model/GRU.yaml:4 hidden_size: 768)
            "C4_encoder_h768": ["--n-items", "10000000", "--loss", "softmax"]}[name]
    old = sys.argv
    sys.argv = [old[0]] + argv
    try:
        a = parse()
    finally:
        sys.argv = old
    cfg = model_config(a, str(device))
    if name.startswith("C4_encoder"):
        cfg.update(model="GRU", hidden_size=768 if name.endswith("h768") else 128)
    torch.manual_seed(2022)
    model = (GRU if name.startswith("C4_encoder") else SASRec)(cfg)
    opt = SparseDenseAdam(model, lr=1e-3, table_mode="lazy_dense")
    model.train()
    steps, warm = 30, 8
    batches = synth_batches(a, a.n_items, device, 5, n_batches=steps + warm + 2)

    def step(b, nxt):
        opt.zero_grad()
        opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
        opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
        model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
        opt.step(late_join=True)

    ncls = _lib.lib.ur_prof_num_classes()
    names = [_lib.lib.ur_prof_class_name(i).decode() for i in range(ncls)]

    def read():
        ms, cnt, work = (C.c_double * ncls)(), (C.c_int64 * ncls)(), (C.c_double * ncls)()
        _lib.check(_lib.lib.ur_prof_read(ms, cnt, work), "ur_prof_read")
        return {names[i]: dict(ms=ms[i], launches=int(cnt[i]), work=work[i]) for i in range(ncls)}

    for i in range(warm // 2):
        step(batches[i], batches[i + 1])
    _lib.lib.ur_prof_set_mask(0xFFFFFFFF)
    _lib.lib.ur_prof_reset()
    _lib.lib.ur_prof_enable(1)
    for i in range(warm // 2, warm):
        step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    _lib.lib.ur_prof_enable(0)
    per = read()
    dom = max((k for k in per if k != "rows_sort"), key=lambda k: per[k]["ms"])    # (the id sort runs on a side stream under the previous step)
    _lib.lib.ur_prof_reset()
    _lib.lib.ur_prof_set_mask(1 << names.index(dom))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        _lib.lib.ur_prof_enable(1 if i % PROF_EVERY == 0 else 0)
        step(batches[warm + i], batches[warm + i + 1])
    t_host = time.perf_counter() - t0      # everything queued: close to dt = the leg is bound by the host's launch rate, not by the device
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.lib.ur_prof_enable(0)
    c = read()[dom]
    _lib.lib.ur_prof_set_mask(0xFFFFFFFF)
    L, d = a.seq_len, a.d
    seqs = torch.stack([b["item_seq"] for b in batches[warm:warm + steps]])
    first = (seqs > 0).int().argmax(-1)
    lens = torch.where((seqs > 0).any(-1), L - first, torch.full_like(first, L))
    frac = float(lens.float().mean() / L) if (name != "C4_encoder" and (d // a.heads) in (4, 8, 16)) else 1.0
    mfma = dom in MFMA_CLASSES
    scale = frac if dom in ("gemm_nt", "gemm_tn", "row_chain") else 1.0     # GEMM work is counted on the padded row count
    ach = c["work"] * scale / max(c["ms"] * 1e-3, 1e-12) / (1e12 if mfma else 1e9)
    peak = round(class_peak_tflops(dom), 1) if mfma else HBM_PEAK_GBPS   # (a class in split arithmetic: the dense bf16 peak / its piece products)
    out = {"workload": f"{cfg['model']} n_items={a.n_items} d={d} L={L} B={a.batch} K={a.negatives} {a.loss}",
           "ms_per_step": round(dt / steps * 1e3, 4), "examples_per_s": round(a.batch * steps / dt, 1),
           "host_enqueue_ms_per_step": round(t_host / steps * 1e3, 4),
           "roofline": {"bound": "mfma" if mfma else "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": peak,
                        "unit": "TFLOP/s" if mfma else "GB/s", "frac": round(ach / peak, 4), "launches": c["launches"],
                        "avg_launch_us": round(c["ms"] * 1e3 / max(1, c["launches"]), 2)},
           "kernel_time_ms_per_step": {k: round(v["ms"] / (warm - warm // 2), 4) for k, v in per.items() if v["launches"]}}
    del model, opt, batches
    torch.cuda.empty_cache()
    return out


def gather_microbench(table, device):
    """North-star gather target: 100M x 128 fp32 table, uniform random ids; HBM-read GB/s = n*(d*4+8)/t."""
    from unirec_amd import ops
    n = 8 * 1024 * 1024
    N, d = table.shape
    idx = torch.randint(1, N, (n,), device=device)
    ops.embedding_gather(table, idx[:1024])
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.embedding_gather(table, idx)
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
        del out
    best, med = min(times), sorted(times)[len(times) // 2]
    read = n * (d * 4 + 8) / (best * 1e-3) / 1e9
    # HBM bytes per launch from the committed rocprofv3 PMC passes of the same two launches (tools/gather_pmc.sh: FETCH_SIZE x 2 per
    # the gfx950 note, + WRITE_SIZE; separate passes, kernel trace only): counters cannot be read from inside this process
    # (the newest profiles/r*_gather_pmc.json whose kernel-source digest is THIS tree's -- tools/gather_pmc.sh writes it -- else none:
    # a summary of other kernels is not quoted)
    import glob
    pmc, pmc_src = {}, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gather_pmc.json")), reverse=True):
        try:
            js = json.load(open(f))
        except Exception:    # noqa: BLE001
            continue
        if js.get("csrc_digest") == csrc_digest():
            pmc, pmc_src = js.get("per_kernel", {}), os.path.basename(f)
            break

    def traffic(tag):
        for k, v in pmc.items():
            if tag in k and "FETCH_SIZE" in v:
                tot = v["TCC_HIT_sum"] + v["TCC_MISS_sum"] if "TCC_HIT_sum" in v else 0
                tlb = v.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) + v.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0)
                return {"hbm_read_bytes": int(2 * v["FETCH_SIZE"] * 1024), "hbm_write_bytes": int(v.get("WRITE_SIZE", 0) * 1024),
                        "l2_hit_rate": round(v["TCC_HIT_sum"] / tot, 3) if tot else None,
                        "l1_tlb_miss_rate": round(v.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / tlb, 3) if tlb else None,
                        "source": f"profiles/{pmc_src} (same launch shape; kernel sources digest {csrc_digest()}: taken on this tree)"}
        return None
    copy = {"bound": "hbm", "kernel": "gather_kernel<int64,32,4> (gather that WRITES the rows back: n*d*4 B of stores compete for HBM; non-temporal loads and stores)",
            "achieved": round(read, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(read / HBM_PEAK_GBPS, 4),
            "read_plus_write_GBps": round(n * (2 * d * 4 + 8) / (best * 1e-3) / 1e9, 1),
            # a write-back gather moves every row twice: its ceiling is the streaming-copy rate the microarchitecture guide measures
            # (6.29 TB/s read + write), i.e. 0.39 of the 8 TB/s spec as READ rate -- the literal ">= 0.70 of the HBM-read roof" is reachable
            # only for a gather that consumes the rows in registers (the fused gather-dot below, what the training path uses)
            "read_plus_write_frac_of_copy_rate": round(n * (2 * d * 4 + 8) / (best * 1e-3) / 1e9 / 6290.0, 4),
            "n_lookups": n, "table_rows": N,
            "ms": round(best, 4), "ms_median": round(med, 4), "median_GBps": round(n * (d * 4 + 8) / (med * 1e-3) / 1e9, 1),
            "algorithmic_read_bytes": n * (d * 4 + 8), "traffic": traffic("gather_kernel") if (N == 100_000_000 and d == 128) else None}
    # the gather the training path actually uses for candidates: fused gather-dot scorer (rows are consumed in
    # registers, one float written per row) -- C3-shaped: G = 1001 candidates per row
    B, G = 4096, 1001
    cfg = ops.loss_cfg(B, G, d, "softmax")
    cfg.loss_type = -1
    user = torch.randn(B, d, device=device)
    ids = torch.randint(1, N, (B, G), device=device)
    ops.gather_dot_loss_fwd(cfg, user, table, ids)
    torch.cuda.synchronize()
    times2 = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gather_dot_loss_fwd(cfg, user, table, ids)
        e1.record()
        e1.synchronize()
        times2.append(e0.elapsed_time(e1))
    best2, med2 = min(times2), sorted(times2)[len(times2) // 2]
    n2 = B * G
    read2 = n2 * (d * 4 + 8) / (best2 * 1e-3) / 1e9
    copy["fused_gather_dot"] = {"bound": "hbm", "kernel": "scorer_loss_fwd_kernel<32,8> (gather + dot, scores only; 8 rows in flight per lane group, non-temporal row loads)",
                                "achieved": round(read2, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(read2 / HBM_PEAK_GBPS, 4), "n_lookups": n2, "table_rows": N, "ms": round(best2, 4),
                                "ms_median": round(med2, 4), "median_GBps": round(n2 * (d * 4 + 8) / (med2 * 1e-3) / 1e9, 1),
                                "frac_median": round(n2 * (d * 4 + 8) / (med2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                "algorithmic_read_bytes": n2 * (d * 4 + 8),
                                "traffic": traffic("scorer_loss_fwd") if (N == 100_000_000 and d == 128) else None}
    return copy


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ladder
    if (world > 1 or a.supervised) and not a.worker:
        # the launcher's process supervises: the benchmark itself runs in a child per rung of the fallback ladder, under a per-phase
        # watchdog -- a hung collective ends in a JSON line with "hang", never in silence (tools/bench_ladder.py)
        argv = [x for x in sys.argv[1:] if x not in ("--worker", "--supervised")]
        if world == 1:      # one rank: the multi-GPU step at world 1, RCCL at world 1 on the native rungs
            argv += ["--sharded-w1"]
            os.environ["UR_NATIVE_W1"] = "1"
        raise SystemExit(bench_ladder.supervise(os.path.abspath(__file__), argv, rank, world))
    if a.dry_worker:
        return bench_ladder.dry_worker(sys.argv[1:])
    phase = bench_ladder.phase
    phase("init")
    if a.loopback > 1:
        assert world == 1, "--loopback runs W rank threads inside ONE process"
        torch.cuda.set_device(local_rank)
        print(json.dumps({"loopback": loopback_leg(a, torch.device("cuda", local_rank), a.loopback)}), flush=True)
        return
    if os.environ.get("UR_BENCH_SHARE_DEVICE") == "1":   # debugging aid only: several ranks on one GPU (1-GPU dev boxes)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(os.environ.get("UR_BENCH_BACKEND", "nccl"),
                                device_id=device if os.environ.get("UR_BENCH_BACKEND", "nccl") == "nccl" else None)

    from unirec_amd import _lib, ops
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.model.sequential.sasrec import SASRec

    # one distinct batch per step (+1 for the lookahead): cycling a few batches would make every looked-up row "touched
    # 8 steps ago", which is not what uniform ids over n_items rows look like (and costs a 7-step lazy-Adam replay per row)
    def make_batches():
        return synth_batches(a, a.n_items, device, 2022 + 7919 * rank, n_batches=min(a.warmup + a.steps + 1, 1024))

    # The batches (their own generator: the order does not change a value) are made BEFORE the model and the optimizer state: those are
    # ~150 GB of device fills, and the warm-up steps then start on a device that has just been busy.  A device left idle for tens of
    # milliseconds runs the next 20-60 steps 2-5 % slower (tools/warm_probe.py: IDLE_MS = 5 / 50 / 500 in front of a 20-step region:
    # +0.01 / +0.035 / +0.07 ms per step); ~30 batches of a dozen tiny launches each are such a gap.
    phase("setup")
    batches = make_batches()
    torch.manual_seed(2022 + rank)
    cfg = model_config(a, str(device))
    selfcheck = None
    if world > 1 or a.sharded_w1:
        # the SAME optimizer Trainer(config, model) builds under torch.distributed (facility/trainer.py -> facility/distributed.py)
        from unirec_amd.facility.distributed import ShardedSparseDenseAdam
        from unirec_amd.sharded import shard_rows
        if world > 1 and not a.no_selfcheck:
            phase("selfcheck")
            try:
                selfcheck = multi_gpu_selfcheck(a, device, rank, world)
            except AssertionError as e:      # a parity failure is reported in the line (and on stderr), the timing still runs
                selfcheck = {"ok": False, "error": str(e)[:500]}
                print(f"[bench] rank {rank}: multi-GPU self-check FAILED: {e}", file=sys.stderr)
                import torch.distributed as dist
                dist.barrier()
        phase("setup")
        torch.manual_seed(2022 + rank)
        model = SASRec(dict(cfg, n_items=shard_rows(a.n_items, world)))     # the model's table IS this rank's shard: the 100 M-row
        opt = ShardedSparseDenseAdam(model, rank, world, lr=1e-3, table_mode=a.table_mode,   # table never exists in one piece
                                     full_rows={"item_embedding": a.n_items})
        model.train()
        info = {"parallelism": f"dp{world} + embedding rows sharded {world}-way (3 fixed-capacity all-to-alls + 1 flat all-reduce per step; "
                               f"transport: {'library RCCL communicators' if opt._native else 'torch.distributed' if world > 1 else 'none (world 1)'}"
                               f"{'; ladder rung: ' + os.environ['UR_BENCH_RUNG'] if os.environ.get('UR_BENCH_RUNG') else ''})"}

        def step_fn(batch, nxt=None):
            return opt.train_step(batch, None if a.no_prefetch else nxt)
    else:
        model = SASRec(cfg)
        opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
        model.train()
        info = {"parallelism": "single"}

        def step_fn(batch, nxt=None):
            opt.zero_grad()
            opt.plan_batch(item_seq=batch["item_seq"], item_id=batch["item_id"])
            if nxt is not None and not a.no_prefetch:
                # input-pipeline lookahead: the NEXT batch's id sort runs on a side stream under this step's compute
                # (every step still issues exactly one plan; the trainer does the same, facility/trainer.py fit())
                opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
            if a.autograd:
                loss, _, _, _ = model(item_id=batch["item_id"], label=batch["label"], item_seq=batch["item_seq"])
                loss.backward()
            else:   # the trainer's default: the same launches in a straight line, no autograd graph (facility/trainer.py)
                loss = model.forward_backward(item_id=batch["item_id"], label=batch["label"], item_seq=batch["item_seq"])
            opt.step(late_join=nxt is not None)   # as Trainer.train_step does: the next step's forward pass joins the side-stream half
            return loss


    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes as C
    ncls = _lib.lib.ur_prof_num_classes()
    names = [_lib.lib.ur_prof_class_name(i).decode() for i in range(ncls)]

    def prof_read():
        ms, cnt, work = (C.c_double * ncls)(), (C.c_int64 * ncls)(), (C.c_double * ncls)()
        _lib.check(_lib.lib.ur_prof_read(ms, cnt, work), "ur_prof_read")
        return {names[i]: dict(ms=ms[i], launches=int(cnt[i]), work=work[i]) for i in range(ncls)}

    # ---- warm-up: W untimed steps.  Every kernel class is bracketed with HIP events here; the per-class breakdown
    # tells which class dominates.  (Bracketing every launch costs ~0.4 ms/step, so it is NOT left on for `value`.)
    # Python's cyclic garbage collector is kept out of the timed region, as `timeit` does: a generation-2 collection of a process that
    # has imported torch takes ~35 ms of HOST time, lands on an arbitrary step (measured: step 22 of one run, 37 of another, none in a
    # third) and, in a 20-step region whose host runs only ~8 ms ahead of the device, shows up as +1.5 ms/step.  The collector is
    # switched off HERE, in front of the warm-up steps, and nothing is collected: 35 ms of idle device between the warm-up steps and the
    # clock cost the first 20 steps +0.04 ms each (tools/warm_probe.py: 0.746 vs 0.725 for the first 20-step region, 0.705 from the second on)
    import gc
    gc.freeze()     # (no collection here either: it would be 35 ms of idle device right in front of the warm-up steps)
    gc.disable()
    _lib.lib.ur_prof_set_mask(0xFFFFFFFF)
    _lib.lib.ur_prof_reset()
    # The W warm-up steps: the first ones carry one-time costs (code-object loads, workspace allocation), the middle ones are bracketed
    # class by class (HIP events around every launch: the per-class breakdown, and which class gets bracketed in the timed region),
    # and the LAST ones run exactly as the timed steps do, right in front of the clock -- reading some hundred event pairs back is
    # milliseconds of idle device right where the clock would start.  (What a short region still pays, and a long one amortises: one
    # pipeline fill and drain -- the side stream's tail of the last step has no next step to hide under -- ~0.15 ms per region, and a
    # slow ramp over the first ~20 steps after any synchronisation: 20 steps after 5 warm-ups 0.615-0.62 ms, after 50 warm-ups 0.60-0.61,
    # 200 steps 0.596.)
    phase("warmup")
    n_tail = min(2, a.warmup // 2)
    n_first = (a.warmup - n_tail) // 2
    n_prof = a.warmup - n_tail - n_first
    loss = None
    for i in range(n_first + n_prof):
        if i == n_first:
            _lib.lib.ur_prof_enable(0 if a.no_prof else 1)
        loss = step_fn(batches[i % len(batches)], batches[(i + 1) % len(batches)])
    barrier()
    _lib.lib.ur_prof_enable(0)
    warm = prof_read()
    # (a host-side check of the copied scalar: torch.isfinite() here was the first use of four elementwise kernels -- 20-100 ms of
    # code-object loading with the device idle, right in front of the timed region, and an idle device runs its next steps slower)
    import math
    if loss is not None and not math.isfinite(float(loss.detach().item())):
        raise SystemExit("non-finite loss in warm-up")
    dom = max(warm, key=lambda k: warm[k]["ms"]) if (not a.no_prof and any(v["launches"] for v in warm.values())) else None
    # ---- timed region: exactly --steps steps.  Only the dominant kernel class is bracketed (HIP events on the launch
    # stream), and only on every PROF_EVERY-th step, so the measurement perturbs `value` by ~1 %.
    _lib.lib.ur_prof_reset()
    if dom is not None:
        _lib.lib.ur_prof_set_mask(1 << names.index(dom))
    for i in range(n_first + n_prof, a.warmup):     # the last warm-up steps: plain, no read-back behind them
        loss = step_fn(batches[i % len(batches)], batches[(i + 1) % len(batches)])
    phase("timed")
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        if dom is not None:
            _lib.lib.ur_prof_enable(1 if i % PROF_EVERY == 0 else 0)
        loss = step_fn(batches[(a.warmup + i) % len(batches)], batches[(a.warmup + i + 1) % len(batches)])
    t_host = time.perf_counter() - t0   # every launch of the region is queued: close to dt = the host's launch rate bounds the step
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    _lib.lib.ur_prof_enable(0)
    phase("post")
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())
    classes = prof_read()
    # ---- after the timed region: the same class once more with the backward's side stream switched off, i.e. launch
    # durations that are not stretched by the weight-gradient GEMMs running concurrently (reported as roofline.isolated)
    iso = None
    if dom is not None and world == 1:
        prev = _lib.lib.ur_sasrec_set_side_stream(0)
        _lib.lib.ur_prof_reset()
        _lib.lib.ur_prof_enable(1)
        for i in range(10):
            step_fn(batches[(a.warmup + i) % len(batches)], None)
        barrier()
        _lib.lib.ur_prof_enable(0)
        _lib.lib.ur_sasrec_set_side_stream(prev)
        iso = prof_read()[dom]
    _lib.lib.ur_prof_set_mask(0xFFFFFFFF)
    collectives = None
    if world > 1 or (a.sharded_w1 and getattr(opt, "_native", False)):   # per-collective device time and bytes to the peers, 10 extra steps after the timed region (all ranks take part; one rank through RCCL: the groups' fixed cost)
        coll_names = ("a2a_ids", "a2a_rows", "a2a_row_grads", "allreduce")
        if opt._native:   # the library's RCCL route: its own profiler classes (events on the stream each group runs on)
            _lib.lib.ur_prof_reset()
            _lib.lib.ur_prof_set_mask(sum(1 << names.index(n) for n in coll_names))
            _lib.lib.ur_prof_enable(1)
        else:
            opt.xchg.profile_start()
        # (with the lookahead batch, as in the timed region: the rows travel a step ahead and the step itself pays the fix-up exchange --
        # torch.distributed route: labels a2a_rows (prefetched, plan stream) / a2a_fix_slots / a2a_fix_rows; library route: both row
        # exchanges are booked on a2a_rows, two calls a step, the slot lists on a2a_ids)
        for i in range(10):
            step_fn(batches[(a.warmup + i) % len(batches)], None if a.no_prefetch else batches[(a.warmup + i + 1) % len(batches)])
        barrier()
        if opt._native:
            _lib.lib.ur_prof_enable(0)
            pr = prof_read()
            _lib.lib.ur_prof_set_mask(0xFFFFFFFF)
            collectives = {n: {"ms_per_step": round(pr[n]["ms"] / 10, 4), "MB_to_peers_per_step": round(pr[n]["work"] / 10 / 1e6, 3),
                               "calls_per_step": pr[n]["launches"] / 10, "route": "library RCCL communicators"} for n in coll_names}
        else:
            prof = opt.xchg.profile_stop()
            collectives = {k: {"ms_per_step": round(v["ms"] / 10, 4), "MB_to_peers_per_step": round(v["MB_to_peers"] / 10, 3),
                               "calls_per_step": v["calls"] / 10, "route": "torch.distributed"} for k, v in prof.items()}
    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.barrier()      # (every rank's worker ends together: rank 0 prints its line in front of this barrier's counterpart below)
        return

    B, L, G, d = a.batch, a.seq_len, a.negatives + 1, a.d
    # fraction of the B*L token slots that hold real tokens (left padding skipped; an all-padding row keeps its L slots)
    seqs = torch.stack([b["item_seq"] for b in batches[a.warmup:a.warmup + a.steps]])
    first = (seqs > 0).int().argmax(-1)
    lens = torch.where((seqs > 0).any(-1), L - first, torch.full_like(first, L))
    valid_frac = float(lens.float().mean() / L) if (L <= 64 and (d // a.heads) in (4, 8, 16)) else 1.0
    ex_per_s = world * B * a.steps / dt
    ms_per_step = dt / a.steps * 1e3
    # dominant kernel class by measured device time inside the timed region
    c = classes[dom] if dom is not None else {"ms": 0}
    if c["ms"] <= 0:
        roof = None   # --no-prof: kernels were not bracketed
    elif dom in ("gemm_nt", "gemm_tn", "row_chain"):
        # the library counts 2*M*N*K with the PADDED row count M = B*L; with padding skipped (the default) only the rows
        # of real tokens are computed, so the algorithmic flops are scaled by the batches' real-token fraction
        achieved = c["work"] * valid_frac / (c["ms"] * 1e-3) / 1e12
        # priced against the ceiling of the pipe the class RUNS on: a class in split arithmetic executes `terms` bf16 MFMA flops per
        # algorithmic (fp32-equivalent) flop, so its ceiling is the dense bf16 peak / terms; the fraction of the fp32-input peak is kept beside it
        peak, terms = class_peak_tflops(dom), class_terms(dom)
        roof = {"bound": "mfma", "kernel": f"{dom} kernels (v_mfma_f32_32x32x2_f32)", "achieved": round(achieved, 2),
                "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                "launches": c["launches"], "avg_launch_us": round(c["ms"] * 1e3 / max(1, c["launches"]), 2),
                "real_token_fraction": round(valid_frac, 4)}
        if terms:
            roof["kernel"] = f"{dom} kernels (v_mfma_f32_32x32x16_bf16 on exactly split fp32 operands, {terms} piece products per product)"
            roof["peak_is"] = f"dense bf16 MFMA peak {MFMA_BF16_PEAK_TFLOPS:.0f} / {terms}: fp32-equivalent TFLOP/s"
            roof["frac_of_fp32_mfma_peak"] = round(achieved / MFMA_F32_PEAK_TFLOPS, 4)
        if iso is not None and iso["ms"] > 0:
            ach_iso = iso["work"] * valid_frac / (iso["ms"] * 1e-3) / 1e12
            roof["isolated"] = {"achieved": round(ach_iso, 2), "frac": round(ach_iso / peak, 4), "launches": iso["launches"],
                                "avg_launch_us": round(iso["ms"] * 1e3 / max(1, iso["launches"]), 2),
                                "note": "same class, 10 extra steps after the timed region with the side stream off (no concurrent dW GEMMs)"}
    else:
        achieved = c["work"] / (c["ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "launches": c["launches"],
                "avg_launch_us": round(c["ms"] * 1e3 / max(1, c["launches"]), 2)}
    # HBM bytes per launch of the dominant class: PMC counters cannot be collected from inside this process, so the
    # figure is the committed rocprofv3 --pmc summary of the SAME workload (profiles/r*_pmc_hbm_traffic.json, the latest:
    # separate FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH x2 correction); null when the summary is absent
    import glob
    digest = csrc_digest()
    pmc, pmc_json, stale = "", None, []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):   # newest first
        try:
            js = json.load(open(f))
        except Exception:    # noqa: BLE001
            continue
        if js.get("csrc_digest") == digest:
            pmc, pmc_json = f, js
            break
        stale.append(os.path.basename(f))
    if roof is not None:
        roof["traffic_source"] = (f"profiles/{os.path.basename(pmc)} (kernel sources digest {digest}: taken on this tree)" if pmc else
                                  f"none: no committed PMC summary was taken on this tree's kernels (digest {digest}); newest other: {stale[:1]}")
    if roof is not None and pmc and a.n_items == 100_000_000 and a.batch == 512:
        per_class = pmc_json.get("per_class", {})
        if dom in per_class and per_class[dom]:
            roof["traffic"] = per_class[dom]["hbm_bytes_per_launch"]
            roof["traffic_unit"] = f"HBM bytes per launch (rocprofv3 PMC, profiles/{os.path.basename(pmc)})"
            # per launch of the SAME kernels the PMC summary counts (the class's main kernel: gemm_tn = the grouped weight-gradient launches,
            # two a step; the deferred reduce_batch launches of the class are a PMC class of their own and carry no operand traffic)
            lps = per_class[dom]["launches"] / max(1, pmc_json.get("steps_profiled", 1))
            roof["algorithmic_bytes_per_launch"] = int(ALGO_BYTES_PER_STEP.get(dom, 0) * valid_frac * 1e6 / max(1.0, lps))
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / max(1, roof["algorithmic_bytes_per_launch"]), 3)
    # the whole step against its two floors: algorithmic flops of the MFMA classes (the launchers' own counts from the bracketed warm-up
    # steps, padded-row counts scaled to the real token rows) at the fp32 MFMA peak, and every kernel's PMC bytes at the rate a
    # streaming copy reaches on this part
    n_prof_steps = max(1, n_prof)
    useful = sum(v["work"] * (valid_frac if k in ("gemm_nt", "gemm_tn", "row_chain") else 1.0) for k, v in warm.items() if k in MFMA_CLASSES) / n_prof_steps / 1e9
    pmc_bytes = pmc_json.get("hbm_bytes_per_step_all_kernels") if (pmc_json and a.n_items == 100_000_000 and a.batch == 512) else None
    step_floor = None
    if useful > 0:
        # every class at the ceiling of the pipe it runs on (split arithmetic: the dense bf16 peak / piece products)
        mfma_floor_ms = sum(v["work"] * (valid_frac if k in ("gemm_nt", "gemm_tn", "row_chain") else 1.0) / class_peak_tflops(k)
                            for k, v in warm.items() if k in MFMA_CLASSES) / n_prof_steps / 1e12 * 1e3
        step_floor = {"useful_gflop": round(useful, 2), "mfma_floor_ms": round(mfma_floor_ms, 4), "frac_of_mfma_floor": round(mfma_floor_ms / ms_per_step, 4),
                      "pmc_bytes": pmc_bytes,
                      "hbm_floor_ms": round(pmc_bytes / (HBM_COPY_GBPS * 1e9) * 1e3, 4) if pmc_bytes else None,
                      "frac_of_hbm_floor": round(pmc_bytes / (HBM_COPY_GBPS * 1e9) * 1e3 / ms_per_step, 4) if pmc_bytes else None,
                      "note": f"useful_gflop = algorithmic flops of the MFMA classes per step (launcher counts, real token rows); floors: every class at its own "
                              f"ceiling ({MFMA_F32_PEAK_TFLOPS} TFLOP/s fp32-input MFMA; split-arithmetic classes {MFMA_BF16_PEAK_TFLOPS:.0f} / piece products) and {HBM_COPY_GBPS / 1e3:.2f} TB/s (streaming-copy rate of this part); pmc_bytes = all kernels of a step, "
                              f"from the same summary as roofline.traffic (null when none was taken on this tree)"}
    emb_bytes_per_example = 8 * (L + G) * d * 4   # fwd read + bwd/opt touched rows (w,m,v,grad)
    out = {
        "metric": "training_examples_per_sec", "value": round(ex_per_s, 1), "unit": "examples/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "host_enqueue_ms_per_step": round(t_host / a.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"SASRec n_items={a.n_items} d={d} seq_len={L} n_layers={a.layers} n_heads={a.heads} inner={a.inner} "
                               f"act=swish, {a.negatives} uniform negatives, {a.loss} loss, per-GPU batch {B}, ids={a.ids}, "
                               f"embedding optimizer={a.table_mode} Adam, dropout={a.dropout:g}",
                   "global_batch": world * B, "seq_len": L, "parallelism": info["parallelism"]},
        "hbm_embedding_GBps_algorithmic": round(ex_per_s * emb_bytes_per_example / 1e9, 2),
        "final_loss": round(final_loss, 6),
        # arithmetic of the dense contractions: every value is fp32 (`dtype`); the weight-gradient products are evaluated either on the
        # fp32-input MFMA or -- the default since round 6 -- as six bf16 piece products of exactly split fp32 operands, fp32-accumulated
        "mfma_arith": mfma_arith_note(),
        "roofline": roof,
        "step_floor": step_floor,
        "kernel_time_ms_per_step_warmup": {k: round(v["ms"] / max(1, n_prof), 4) for k, v in warm.items() if v["launches"]},
        # every MFMA-bound class, from the warm-up steps where all classes are bracketed (same definition as `roofline`: algorithmic flops
        # of the real token rows / device time of the class, in situ -- the weight-gradient GEMMs share the CUs with the main stream)
        "mfma_classes_warmup": {k: {"TFLOPs": round(v["work"] * (valid_frac if k in ("gemm_nt", "gemm_tn", "row_chain") else 1.0) / (v["ms"] * 1e-3) / 1e12, 2),
                                    "peak": round(class_peak_tflops(k), 1),
                                    "frac": round(v["work"] * (valid_frac if k in ("gemm_nt", "gemm_tn", "row_chain") else 1.0) / (v["ms"] * 1e-3) / 1e12 / class_peak_tflops(k), 4),
                                    "launches_per_step": round(v["launches"] / max(1, n_prof), 1),
                                    **({"frac_of_fp32_mfma_peak": round(v["work"] * valid_frac / (v["ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
                                       if class_terms(k) else {})}
                                for k, v in warm.items() if k in MFMA_CLASSES and v["ms"] > 0 and v["work"] > 0},
    }
    if world == 1 and a.sharded_w1:
        from unirec_amd import ops as _ops
        out["n_ranks"], out["rccl_ranks"] = 1, _ops.comm_count()
        if collectives is not None:
            out["collectives"] = collectives
    if world > 1:
        import torch.distributed as dist
        out["n_ranks"] = dist.get_world_size()
        from unirec_amd import ops as _ops
        out["rccl_ranks"] = _ops.comm_count()     # ncclCommCount of the library's own communicators (0: the torch.distributed route carried the exchange)
        out["backend"] = dist.get_backend()
        out["selfcheck"] = selfcheck if selfcheck is not None else "skipped (--no-selfcheck)"
        out["collectives"] = collectives
    if world == 1 and not a.no_extra_legs and not a.autograd:
        n_leg = max(10, min(a.steps, 50))
        out["e2e"] = e2e_leg(a, model, opt, device, n_leg)
        if a.dropout == 0.0 and not a.sharded_w1:
            out["dropout_0p5"] = dict(variant_leg(a, model, opt, step_fn, device, n_leg, dropout=0.5),
                                      note="hidden_dropout_prob = attn_dropout_prob = 0.5: the reference's SASRec.yaml default")
        if a.ids == "uniform":
            out["zipf_ids"] = dict(variant_leg(a, model, opt, step_fn, device, n_leg, ids="zipf"), note="item ids ~ Zipf(1.0) instead of uniform")
        out["steady_state"] = steady_state_leg(a, opt, step_fn, batches, n_leg, barrier)      # (last: it ages the optimizer state)
    if world == 1 and not a.no_gather_bench:
        out["gather_roofline"] = gather_microbench(model.item_embedding.weight.data, device)
    if world == 1:      # (the legs below build models of their own: the headline's 100 M-row table and its optimizer state go first)
        del model, opt
        torch.cuda.empty_cache()
    if world == 1 and not a.no_extra_legs and not a.autograd and not a.sharded_w1 and a.dropout == 0.0:
        out["trainer_fit"] = trainer_fit_leg(a, device, steps=min(1000, max(200, a.steps)))      # (as many steps as the headline: the same table ageing; after the headline's model is gone: a second 100 M-row table + state)
    if world == 1 and a.all_configs:
        out["other_configs"] = {n: other_config(n, device) for n in ("C2", "C3", "C4_encoder", "C4_encoder_h768")}
        out["other_configs"]["C3"]["e2e"] = c3_e2e_leg(device)
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a)
    print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


if __name__ == "__main__":
    main()
