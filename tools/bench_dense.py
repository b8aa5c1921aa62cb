"""Micro-benchmark of the wide dense-layer kernels at DCN's shapes (run under rocprofv3
--kernel-trace --stats; the kernel durations are the measurement, not the host loop)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recman_amd import ops  # noqa: E402

B, FD, Dn, H = 131072, 416, 13, 400
dev = "cuda"
xe, xd = torch.randn(B, FD, device=dev), torch.randn(B, Dn, device=dev)
W1, W2 = torch.randn(FD + Dn, H, device=dev) * 0.05, torch.randn(H, H, device=dev) * 0.05
b1 = torch.zeros(H, device=dev)
h1, h2, dh = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev), torch.randn(B, H, device=dev)
dxe = torch.empty(B, FD, device=dev)
dW1, dW2 = torch.empty(FD + Dn, H, device=dev), torch.empty(H, H, device=dev)
fws = torch.empty(ops.dense_filter_workspace(FD + Dn, 448), device=dev)
wws = torch.empty(ops.dense_wgrad_workspace(FD + Dn, H, B), device=dev)
x = torch.cat([xe, xd], 1)
for it in range(12):
    ops.dense_fwd(xe, xd, W1, h1, fws, bias=b1, act="relu")                       # layer 0
    ops.dense_fwd(h1, None, W2, h2, fws, bias=b1, act="relu")                     # layer 1
    ops.dense_fwd(dh, None, W2, h2, fws, transposed=True, epilogue=ops.DENSE_MUL_ACTGRAD, act="relu", aux1=h1)
    ops.dense_fwd(dh, None, W1[:FD], dxe, fws, transposed=True, epilogue=ops.DENSE_ADD)  # dX (N=416)
    ops.dense_wgrad(xe, xd, dh, dW1, wws)
    ops.dense_wgrad(h1, None, dh, dW2, wws)
    torch.mm(x, W1, out=h1)            # hipBLASLt for comparison
    torch.mm(x.t(), dh, out=dW1)
torch.cuda.synchronize()
print("ok")
