"""Experiment: moments inside a 256-byte table row (rows [R, 64], mom = rows[:, 32:]) against separate arrays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import ops

B, F, V, D = 65536, 26, 1000001, 16
R = F * V
g = torch.Generator().manual_seed(0)
idx = torch.randint(0, V, (B, F), generator=g).cuda()
foff = (torch.arange(F) * V).cuda()
rows = torch.zeros(R, 64, device="cuda")
mom = rows.view(-1)[32:32 + R * 32].view(R, 32)   # (the experimental build strides it by 64 floats)
d_rows = torch.randn(B, F, D, device="cuda")
gb = torch.randn(B, device="cuda")
ws = torch.zeros(ops.sparse_optimizer_workspace(B * F), dtype=torch.uint8, device="cuda")
t = [1]
def step():
    ops.sparse_optimizer_step(idx, foff, d_rows, rows, mom, ws, t[0], "adam", 0.01, g_bias=gb, g_lin=gb,
                              max_field_rows=V)
    t[0] += 1
for _ in range(5):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30):
    step()
b.record()
torch.cuda.synchronize()
print(f"256-byte rows with the moments inside: step {a.elapsed_time(b) / 30 * 1e3:.1f} us")
