#!/usr/bin/env python
"""Times rm_cross_fwd / rm_cross_bwd alone at the config-4 shape (hipEvents)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recman_amd import ops
B, FD, Dn, L = 131072, 416, 13, 6
dev = "cuda"
d = FD + Dn
xe = torch.randn(B, FD, device=dev); xd = torch.randn(B, Dn, device=dev)
w = torch.randn(L, d, device=dev) * 0.05; b = torch.randn(L, d, device=dev) * 0.05; wo = torch.randn(d, device=dev) * 0.05
logit = torch.empty(B, device=dev); s = torch.empty(B, ops.cross_p_ld(L), device=dev); g = torch.randn(B, device=dev)
dxin = torch.randn(B, FD, device=dev); dxe = torch.empty(B, FD, device=dev); coef = torch.empty(B, 2 * L + 2, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, c in ev:
        a.record(); fn(); c.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(c) for a, c in ev) / n * 1e3
t = timeit(lambda: ops.cross_fwd(xe, xd, w, b, wo, logit, s))
print(f"cross_fwd {t:7.1f} us  {B*(d*4+4+4*(L+1))/t/1e3:7.1f} GB/s")
t = timeit(lambda: ops.cross_bwd(w, b, wo, g, s, dxe, coef, dx_in_e=dxin))
print(f"cross_bwd {t:7.1f} us  {B*(2*FD*4+4*(2*L+2)+4*(L+1)+4)/t/1e3:7.1f} GB/s")
