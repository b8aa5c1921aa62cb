"""Times rm_cin_layer_bwd (split kernels) at configs[2] layer 1: B = 65536, m = 26, H = 64, N = 128, D = 16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import ops
B, m, H, N, D = 65536, 26, 64, 128, 16
g = torch.Generator(device="cuda").manual_seed(0)
X0 = torch.randn(B, m, D, generator=g, device="cuda")
Xk = torch.randn(B, H, D, generator=g, device="cuda")
W = torch.randn(m * H, N, generator=g, device="cuda") * 0.05
out = torch.randn(B, N, D, generator=g, device="cuda")
gv = torch.randn(B, generator=g, device="cuda")
dh = torch.randn(B, N // 2, D, generator=g, device="cuda")
cw = torch.randn(N // 2, generator=g, device="cuda")
dX0, dXk = torch.zeros(B, m, D, device="cuda"), torch.empty(B, H, D, device="cuda")
dW, db = torch.empty(m * H, N, device="cuda"), torch.empty(N, device="cuda")
ws = torch.empty(ops.cin_bwd_workspace(B, m, H, N, D), device="cuda")
def f():
    ops.cin_layer_bwd(X0, Xk, H, W, "leaky_relu", out, gv, dX0, dW, db, ws, d_hidden=dh, cin_w_direct=cw,
                      pool_from=N // 2, accumulate_dx0=True, dXk=dXk, split=True)
for _ in range(2):
    f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    f()
b.record()
torch.cuda.synchronize()
print(f"layer-1 backward (dm + dx6 + dw6 + reduces): {a.elapsed_time(b) / 5 * 1e3:.0f} us")
