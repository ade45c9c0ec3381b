import torch, time
dev="cuda:0"
def t(fn,n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
M=25600
print("TN (dW = dY^T X): T=25600")
for R,C in [(128,512),(512,128),(128,128),(384,128),(256,128)]:
    dY=torch.randn(M,R,device=dev); X=torch.randn(M,C,device=dev)
    us=t(lambda: torch.mm(dY.t(),X))
    print(f"  R={R} C={C}: {us:7.1f} us {2*M*R*C/us/1e6:6.1f} TF/s")
print("NT (Y = X W^T): M=25600")
for N,K in [(384,128),(512,128),(128,512),(128,128),(128,384),(256,128)]:
    X=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)
    us=t(lambda: torch.mm(X,W.t()))
    print(f"  N={N} K={K}: {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s")
