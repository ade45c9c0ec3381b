import faulthandler, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unirec_amd import ops
dev = torch.device('cuda:0')
ops.id_guard_reset()
print('reset ok', flush=True)
ids = torch.tensor([1, 5, 1000, 3], dtype=torch.int64, device=dev)
pl = ops.rows_plan(None, ids, 1000)
torch.cuda.synchronize()
print('plan ok', pl.uniq_idx[:4].tolist(), int(pl.n_uniq), flush=True)
try:
    ops.id_guard_check()
    print('NOT RAISED', flush=True)
except IndexError as e:
    print('raised:', e, flush=True)
ops.id_guard_reset()
ops.id_guard_check()
print('clear ok', flush=True)
