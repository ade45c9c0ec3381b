import os, sys, torch
sys.path.insert(0, "/root/repo")
from unirec_amd import ops
dev = torch.device("cuda:0")
B, L, d, I, H, nl = 512, 50, 128, 512, 16, 2
cfg = ops.sasrec_cfg(B, L, d, H, I, nl, "swish", True, 1e-10, last_only=1, skip_padding=1, p_hidden=0.0, p_attn=0.0, drop_seed=7, drop_step=3)
offs, total = ops.sasrec_param_layout(cfg)
g = torch.Generator(device=dev).manual_seed(0)
N = 20000
ws = ops.sasrec_workspace(cfg, dev); ws.zero_()
outs = []
for it in range(40):
    dense = torch.randn(total, device=dev, generator=g) * 0.08
    table = torch.randn(N, d, device=dev, generator=g) * 0.1
    seq = torch.randint(1, N, (B, L), device=dev, generator=g, dtype=torch.int32)
    du = torch.randn(B, d, device=dev, generator=g)
    ue = ops.sasrec_fwd(cfg, table, dense, seq, ws).clone()
    dg, dr = ops.sasrec_bwd(cfg, table, dense, seq, du, ws)
    torch.cuda.synchronize()
    outs.append((ue.cpu(), dg.clone().cpu(), dr.clone().cpu()))
torch.save(outs, sys.argv[1])
