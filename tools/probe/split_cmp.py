import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
worst = [0.0, 0.0, 0.0]
for it, (x, y) in enumerate(zip(a, b)):
    for k in range(3):
        s = float(y[k].abs().max())
        e = float((x[k] - y[k]).abs().max()) / max(s, 1e-30)
        if e > 1e-5: print("step", it, "tensor", k, "rel err", e)
        worst[k] = max(worst[k], e)
print("worst relative differences (user_emb, dense_grad, d_rows):", worst)
