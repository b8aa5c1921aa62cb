#!/usr/bin/env python
"""Micro-benchmark of rm_embed_fwd table layouts at the BASELINE config-2 shape.
Variants: separate tables (3 random accesses per lookup), rows only, fused rows
[D | bias | lin | pad] at 80 B and 128 B strides."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recman_amd import ops

B, F, D, V, Dn = 65536, 26, 16, 1_000_001, 13
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
R = F * V
idx = torch.randint(0, V, (B, F), generator=g, device=dev)
dense = torch.randn(B, Dn, device=dev)
foff = (torch.arange(F, device=dev) * V)
E = torch.empty(B, F, D, device=dev); S = torch.empty(B, D, device=dev)
fm = torch.empty(B, device=dev); lin = torch.empty(B, device=dev)
lwd = torch.randn(Dn, device=dev); lw0 = torch.randn(1, device=dev)


def timeit(fn, n=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / n * 1e3

alg = B * (F * 8 + 2 * F * 4 * D + F * 4 + 4 * D + 4 + F * 4 + Dn * 4 + 4)
res = {}
table = torch.randn(R, D, device=dev) * 0.01
bias = torch.randn(R, device=dev) * 0.01
linw = torch.randn(R, device=dev) * 0.01
res["separate tables (rows+bias+lin)"] = timeit(lambda: ops.embed_fwd(idx, table, foff, bias_table=bias, lin_w=linw, lin_off=foff, lin_w_dense=lwd, lin_w0=lw0, dense=dense, E=E, fm_sum=S, fm_logit=fm, lin_logit=lin))
res["rows only (no bias/lin, E+S out)"] = timeit(lambda: ops.embed_fwd(idx, table, foff, E=E, fm_sum=S))
res["rows only, E not written"] = timeit(lambda: ops.embed_fwd(idx, table, foff, fm_sum=S))
del table, bias, linw
for ld in (20, 32):
    t = torch.randn(R, ld, device=dev) * 0.01
    flat = t.view(-1)
    res[f"fused rows ld={ld} ({ld*4} B)"] = timeit(lambda: ops.embed_fwd(idx, t, foff, table_ld=ld, D=D, bias_table=flat[D:], bias_ld=ld, lin_w=flat[D + 1:], lin_ld=ld, lin_off=foff, lin_w_dense=lwd, lin_w0=lw0, dense=dense, E=E, fm_sum=S, fm_logit=fm, lin_logit=lin))
    res[f"fused rows ld={ld}, rows only"] = timeit(lambda: ops.embed_fwd(idx, t, foff, table_ld=ld, D=D, E=E, fm_sum=S))
    del t, flat
# FUSED ENGINE LAYOUT: ld = 2*D with bias/lin in the row -> embed_fwd_fused_kernel
t = torch.randn(R, 32, device=dev) * 0.01
res["engine layout (fused kernel, ld=32)"] = timeit(lambda: ops.embed_fwd(idx, t, foff, table_ld=32, D=D, bias_col=D, lin_col=D + 1, lin_w_dense=lwd, lin_w0=lw0, dense=dense, E=E, fm_sum=S, fm_logit=fm, lin_logit=lin))
res["engine layout, no E write"] = timeit(lambda: ops.embed_fwd(idx, t, foff, table_ld=32, D=D, bias_col=D, lin_col=D + 1, lin_w_dense=lwd, lin_w0=lw0, dense=dense, fm_sum=S, fm_logit=fm, lin_logit=lin))
del t
for k, v in res.items():
    print(f"{k:45s} {v:8.1f} us   {alg / v / 1e3:7.1f} GB/s (config-2 algorithmic bytes)")
