"""Watchdog + fallback ladder for ``bench.py --gpus N`` (N > 1): the bench can no longer return nothing.

The process the launcher starts (one per rank: ``python -m torch.distributed.run ... bench.py --gpus N``) never touches the GPU.  It is
a SUPERVISOR: it runs the actual benchmark in a child process (``bench.py ... --worker``, own session, own rendezvous port), watches the
child's phase marks (a one-line status file the worker rewrites at every phase: init / setup / selfcheck / warmup / timed / post) and
kills the child's process group when a phase outlives its limit -- a hung collective cannot be recovered in-process, a hung process can
be replaced.  The supervisors of one launch agree through a scratch directory (one node: /tmp): a rung fails for all ranks as soon as
it fails for one, and they walk the ladder together:

    rung 0  native-2comm-prefetch   the library's two RCCL communicators on three streams, rows fetched a step ahead (the default)
    rung 1  native-2comm            the same, rows inside the step (UR_PREFETCH_ROWS=0): no fix-up exchange, no plan-stream row traffic
    rung 2  native-1comm-1stream    ONE communicator, everything on the caller's stream (UR_COMM_SINGLE=1, UR_DENSE_SIDE=0, no lookahead):
                                    the conservative shape of the reference's own DDP step (unirec/facility/trainer.py:67, 346-349)
    rung 3  torch.distributed       no library communicator at all: packed blocks through torch.distributed collectives

Rank 0's supervisor prints the ONE JSON line: the first successful rung's line with ``ladder`` (what was tried, which phase hung) and
the rung in ``config.parallelism``; when every rung fails, a line with ``value`` 0 and ``"hang": <phase>`` -- never silence.

Test hooks (tests/test_bench_ladder.py): UR_BENCH_TEST_HANG / UR_BENCH_TEST_KILL = "<rank>:<phase>:<rung>[,<rung>...]" make that rank's
worker sleep forever / die at that phase on those rungs; UR_BENCH_TIMEOUT_SCALE scales every limit; ``--dry-worker`` walks the phases
over gloo with no GPU work (the supervisor logic on a CPU box)."""
import json
import os
import signal
import subprocess
import sys
import tempfile
import time

RUNGS = [
    ("native-2comm-prefetch", {}, []),
    ("native-2comm", {"UR_PREFETCH_ROWS": "0"}, []),
    ("native-1comm-1stream", {"UR_PREFETCH_ROWS": "0", "UR_COMM_SINGLE": "1", "UR_DENSE_SIDE": "0"}, ["--no-prefetch"]),
    ("torch.distributed", {"UR_PREFETCH_ROWS": "0", "UR_NATIVE_TRANSPORT": "0"}, []),
]
# seconds a phase may take before the rank group is aborted (x UR_BENCH_TIMEOUT_SCALE).  Generous: the first `import torch` on a fresh box
# pages the image in for 1-2 minutes, a 100 M-row table plus optimizer state is ~150 GB of fills, RCCL's first communicator takes seconds
LIMITS = {"start": 300.0, "init": 180.0, "setup": 300.0, "selfcheck": 240.0, "warmup": 150.0, "timed": 150.0, "post": 300.0}
PHASES = list(LIMITS)
# ... and seconds a whole RUNG may take, whatever its phases do (x UR_BENCH_TIMEOUT_SCALE): four rungs that all hang end within
# 480 + 280 + 180 + 140 s = 18 min, under the driver's 30-minute limit with a cold node's first `import torch` inside rung 0 (the later
# rungs start with the image paged in and the file cache warm: a healthy rung takes 2-3 min on a cold node, ~1 min after)
RUNG_BUDGET = [480.0, 280.0, 180.0, 140.0]


# ---------------------------------------------------------------------------------------------------------------- worker side
def phase(name):
    """called by the worker at the head of every phase: rewrite the status file, serve the test hooks"""
    path = os.environ.get("UR_BENCH_STATUS")
    if path:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            f.write(f"{name} {time.time():.3f}\n")
        os.replace(tmp, path)
    rank, rung = os.environ.get("RANK", "0"), os.environ.get("UR_BENCH_RUNG_INDEX", "0")
    for var, act in (("UR_BENCH_TEST_HANG", "hang"), ("UR_BENCH_TEST_KILL", "kill")):
        spec = os.environ.get(var)
        if not spec:
            continue
        r, ph, rungs = spec.split(":")
        if r == rank and ph == name and rung in rungs.split(","):
            if act == "kill":
                os._exit(17)
            while True:           # a hang: this rank never arrives at the next collective
                time.sleep(1.0)


def dry_worker(argv):
    """the phases over gloo, no GPU: what the supervisor tests drive on a CPU box (every phase ends in a barrier, as the real ones do)"""
    import torch
    import torch.distributed as dist
    phase("init")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for name in ("setup", "selfcheck", "warmup", "timed", "post"):
        phase(name)
        t = torch.ones(1)
        dist.all_reduce(t)
        assert int(t) == world
    if rank == 0:
        print(json.dumps({"metric": "training_examples_per_sec", "value": 1.0, "unit": "examples/s", "n_gpus": world, "dry": True,
                          "config": {"parallelism": f"dry run ({os.environ.get('UR_BENCH_RUNG', '?')})"}}))
    dist.barrier()
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------- supervisor side
def _read_status(path):
    try:
        with open(path) as f:
            name, t = f.read().split()
        return name, float(t)
    except Exception:    # noqa: BLE001  (not written yet / being replaced)
        return None


def _kill_group(proc):
    """the child runs in its own session: signal exactly that process group (never a pattern)"""
    if proc.poll() is not None:
        return
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)
        except ProcessLookupError:
            return
        for _ in range(20):
            if proc.poll() is not None:
                return
            time.sleep(0.1)


def supervise(bench_py, argv, rank, world, out=sys.stdout):
    scale = float(os.environ.get("UR_BENCH_TIMEOUT_SCALE", "1"))
    base_port = int(os.environ.get("MASTER_PORT", "29500"))
    launch_id = f"{base_port}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}_{os.getppid()}"
    root = os.path.join(tempfile.gettempdir(), f"ur_bench_ladder_{launch_id}")
    os.makedirs(root, exist_ok=True)
    first = int(os.environ.get("UR_BENCH_FIRST_RUNG", "0"))
    ladder, line = [], None
    for k, (name, env_add, arg_add) in enumerate(RUNGS):
        if k < first:
            continue
        status = os.path.join(root, f"rung{k}.rank{rank}.status")
        done = os.path.join(root, f"rung{k}.rank{rank}.done")
        failed_flag = os.path.join(root, f"rung{k}.failed")
        verdict = os.path.join(root, f"rung{k}.verdict")
        env = dict(os.environ, **env_add)
        # the workers' own rendezvous port: rank 0's supervisor asks the kernel for a free one and publishes it (a fixed offset from the
        # launcher's port can be taken -- the rung would then fail in `init` for no fault of its own)
        port_file = os.path.join(root, f"rung{k}.port")
        port = base_port + 101 + 7 * k
        if rank == 0:
            try:
                import socket
                with socket.socket() as sk:
                    sk.bind(("", 0))
                    port = sk.getsockname()[1]
            except OSError:
                pass
            with open(port_file + ".tmp", "w") as f:
                f.write(str(port))
            os.replace(port_file + ".tmp", port_file)
        else:
            deadline = time.time() + 60.0
            while not os.path.exists(port_file) and time.time() < deadline:
                time.sleep(0.05)
            try:
                with open(port_file) as f:
                    port = int(f.read().strip())
            except (OSError, ValueError):
                pass
        env.update(MASTER_PORT=str(port), UR_BENCH_STATUS=status, UR_BENCH_RUNG=name, UR_BENCH_RUNG_INDEX=str(k),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # the children rendezvous among themselves (rank 0's child hosts the store)
        so_path = os.path.join(root, f"rung{k}.rank{rank}.stdout")
        with open(so_path, "w") as so:
            proc = subprocess.Popen([sys.executable, bench_py] + argv + arg_add + ["--worker"], env=env, stdout=so, stderr=sys.stderr,
                                    start_new_session=True)
            t_start, result = time.time(), None
            while result is None:
                time.sleep(0.2)
                st = _read_status(status) or ("start", t_start)
                rc = proc.poll()
                if rc is not None:
                    result = ("ok", st[0]) if rc == 0 else ("exit %d" % rc, st[0])
                elif os.path.exists(failed_flag):
                    result = ("peer failed", st[0])
                elif time.time() - st[1] > LIMITS.get(st[0], 300.0) * scale:
                    result = ("hang", st[0])
                elif time.time() - t_start > RUNG_BUDGET[min(k, len(RUNG_BUDGET) - 1)] * scale:
                    result = ("over the rung's budget", st[0])
            if result[0] != "ok":
                try:
                    with open(failed_flag, "a") as f:
                        f.write(f"rank {rank}: {result[0]} in {result[1]}\n")
                except OSError:
                    pass
                _kill_group(proc)
        with open(done, "w") as f:
            f.write(f"{result[0]}|{result[1]}\n")
        # rank 0 decides for everyone: the rung counts only if EVERY rank's worker finished
        if rank == 0:
            ok, deadline, seen = result[0] == "ok", time.time() + 60.0 * max(1.0, scale), {}
            while len(seen) < world and time.time() < deadline:
                for r in range(world):
                    p = os.path.join(root, f"rung{k}.rank{r}.done")
                    if r not in seen and os.path.exists(p):
                        with open(p) as f:
                            seen[r] = f.read().strip()
                time.sleep(0.1)
            ok = ok and len(seen) == world and all(v.startswith("ok") for v in seen.values())
            with open(verdict + ".tmp", "w") as f:
                f.write("ok" if ok else "failed")
            os.replace(verdict + ".tmp", verdict)
            who = {r: v for r, v in seen.items() if not v.startswith("ok")}
            ladder.append({"rung": name, "ok": ok, "seconds": round(time.time() - t_start, 1), **({"failed": who} if who else {})})
            # evidence after EVERY rung, in case the launch is killed before the ladder ends: a marked line (not the JSON line the
            # contract asks for -- that one comes last, once) on both streams
            note = "[bench-ladder partial] " + json.dumps({"n_gpus": world, "ladder": ladder})
            print(note, file=out, flush=True)
            print(note, file=sys.stderr, flush=True)
        else:
            deadline = time.time() + 90.0 * max(1.0, scale)
            while not os.path.exists(verdict) and time.time() < deadline:
                time.sleep(0.1)
            ok = os.path.exists(verdict) and open(verdict).read().strip() == "ok"
        if ok:
            if rank == 0:
                with open(so_path) as f:
                    lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
                line = json.loads(lines[-1]) if lines else None
            break
    if rank != 0:
        return 0
    if line is None:
        last = ladder[-1] if ladder else {}
        hang = next(iter(last.get("failed", {}).values()), "unknown|unknown").split("|")[-1] if last else "unknown"
        line = {"metric": "training_examples_per_sec", "value": 0.0, "unit": "examples/s", "n_gpus": world, "steps": 0, "warmup": 0,
                "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "SASRec (no rung of the ladder completed)", "parallelism": "none"}, "hang": hang}
    line["ladder"] = ladder
    line.setdefault("config", {})["rung"] = next((e["rung"] for e in ladder if e["ok"]), None)
    print(json.dumps(line), file=out, flush=True)
    return 0
