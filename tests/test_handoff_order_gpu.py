"""Cross-workgroup hand-offs ("the workgroup that arrives last finishes the job": the fused scorer's batch loss / NaN guard, the split
last-row chain kernels' partial tiles) under arrival skew.

The arrival counter is incremented with a RELEASE-ACQUIRE device-scope RMW (csrc/common.h: ur_arrive) -- the form the HIP memory model
recognises; the data travels in device-scope RMWs whose old values have returned before the arrival.  UR_TEST=arrival_skew_us=k delays every
workgroup's arrival by (hash of the workgroup) % k microseconds, so the last arriver -- and the XCD it sits on -- changes from launch to
launch while the partial buffers still hold the previous launch's values.  Every process must produce the same bits: the summation
order is fixed (index order), so any stale read shows up as a different bit pattern."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
from unirec_amd import _lib, ops
dev = torch.device("cuda:0")
h = hashlib.sha256()
# ---- fused scorer + loss: B workgroups hand (loss, count) to the last arriver, 60 launches on ONE set of buffers
g = torch.Generator(device=dev).manual_seed(1)
B, G, d, N = 512, 5, 128, 50000
table = torch.randn(N, d, device=dev, generator=g) * 0.3
for it in range(60):
    ue = torch.randn(B, d, device=dev, generator=g)
    ids = torch.randint(1, N, (B, G), device=dev, generator=g)
    cfg = ops.loss_cfg(B, G, d, "bpr", 1.0, -1.0)
    scores, loss_out, coef, d_user, _ = ops.gather_dot_loss_fwd_bwd(cfg, ue, table, ids, None)
    for t in (scores, loss_out[:3], coef, d_user):      # (loss_out[3] is never written)
        h.update(t.detach().cpu().numpy().tobytes())
# ---- split last-row chain kernels (chain mask 57: the last-row layer through the split kernels, not lastrow.hip)
B, L, d, I, H, nl, N = 512, 50, 128, 512, 16, 2, 20000
cfg = ops.sasrec_cfg(B, L, d, H, I, nl, "swish", True, 1e-10, last_only=1, skip_padding=1, p_hidden=0.0, p_attn=0.0, drop_seed=7, drop_step=3)
_, total = ops.sasrec_param_layout(cfg)
_lib.lib.ur_sasrec_set_chain(57)
ws = ops.sasrec_workspace(cfg, dev)
ws.zero_()
for it in range(20):
    dense = torch.randn(total, device=dev, generator=g) * 0.08
    tab = torch.randn(N, d, device=dev, generator=g) * 0.1
    seq = torch.randint(1, N, (B, L), device=dev, generator=g, dtype=torch.int32)
    du = torch.randn(B, d, device=dev, generator=g)
    ue = ops.sasrec_fwd(cfg, tab, dense, seq, ws).clone()
    dg, dr = ops.sasrec_bwd(cfg, tab, dense, seq, du, ws)
    for t in (ue, dg, dr):
        h.update(t.detach().cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
''' % ROOT


def _run(**env):
    e = dict(os.environ)
    e.pop("UR_TEST", None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1]


@pytest.mark.gpu
def test_handoffs_give_the_same_bits_under_arrival_skew():
    base = _run()
    assert _run(UR_TEST="arrival_skew_us=30") == base                  # the last arriver spread over the XCDs
    assert _run(UR_TEST="arrival_skew_us=200") == base
