"""GRU encoder alone at H = 768 (B = 512, L = 50, d = 128): host enqueue time vs device time of ur_gru_fwd + ur_gru_bwd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import ops
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 768
B, L, d, N = 512, 50, 128, 100000
cfg = ops.gru_cfg(B, L, d, H)
_, total = ops.gru_param_layout(cfg)
g = torch.Generator(device=dev).manual_seed(0)
dense = torch.randn(total, device=dev, generator=g) * 0.03
table = torch.randn(N, d, device=dev, generator=g) * 0.1
seq = torch.randint(1, N, (B, L), device=dev, generator=g, dtype=torch.int32)
du = torch.randn(B, d, device=dev, generator=g)
ws = ops.gru_workspace(cfg, dev)
for _ in range(3):
    ue = ops.gru_fwd(cfg, table, dense, seq, ws); ops.gru_bwd(cfg, table, dense, seq, du, ws)
torch.cuda.synchronize()
for name, fn in (("fwd", lambda: ops.gru_fwd(cfg, table, dense, seq, ws)), ("bwd", lambda: ops.gru_bwd(cfg, table, dense, seq, du, ws))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"H={H} {name}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms, device {e0.elapsed_time(e1) / n:.3f} ms")
