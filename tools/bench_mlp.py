#!/usr/bin/env python
"""Times rm_mlp_fwd / rm_mlp_bwd alone at the config-2 shape (hipEvents)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recman_amd import ops
B, FD, Dn, H = 65536, 416, 13, (32, 32)
dev = "cuda"
xe = torch.randn(B, FD, device=dev); xd = torch.randn(B, Dn, device=dev)
dims = [FD + Dn] + list(H)
Ws = [torch.randn(dims[i], dims[i + 1], device=dev) * 0.1 for i in range(len(H))]
bs = [torch.randn(h, device=dev) * 0.1 for h in H]
wo = torch.randn(H[-1], device=dev); w0 = torch.randn(1, device=dev)
hb = [torch.empty(B, 32, device=dev) for _ in H]; out = torch.empty(B, device=dev)
g = torch.randn(B, device=dev); S = torch.randn(B, 16, device=dev)
d_rows = torch.empty(B, FD, device=dev); dh = [torch.empty(B, 32, device=dev) for _ in H]
dW = [torch.empty_like(W) for W in Ws]; ws = torch.empty(ops.mlp_bwd_workspace(FD, Dn), device=dev)
db = [torch.empty(h, device=dev) for h in H]; dwo = torch.empty(H[-1], device=dev); dw0 = torch.empty(1, device=dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / n * 1e3
print("mlp_fwd us", round(timeit(lambda: ops.mlp_fwd(xe, xd, Ws, bs, wo, w0, "relu", hb, out)), 1))
print("mlp_bwd (+small grads, reduces) us", round(timeit(lambda: ops.mlp_bwd(xe, xd, Ws, wo, "relu", g, hb, d_rows, dh, dW, ws, fm_sum=S, db=db, d_w_out=dwo, d_w0_out=dw0)), 1))
