"""Times rm_sparse_optimizer_prepare (keys + sort) and the whole step with the field-segmented sort against the
one sort over all pairs, at BASELINE configs[1] sizes (B = 65536, 26 fields of 1,000,001 rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import ops

B, F, V, D = 65536, 26, 1000001, 16
R = F * V
g = torch.Generator().manual_seed(0)
idx = torch.randint(0, V, (B, F), generator=g).cuda()
foff = (torch.arange(F) * V).cuda()
rows = torch.randn(R, 32, device="cuda")
rows[:, D + 2: D + 6] = 0
mom = torch.zeros(R, 2 * D, device="cuda")
d_rows = torch.randn(B, F, D, device="cuda")
gb = torch.randn(B, device="cuda")
ws = torch.zeros(ops.sparse_optimizer_workspace(B * F), dtype=torch.uint8, device="cuda")


def timed(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for mfr in (0, V):
    tp = timed(lambda: ops.sparse_optimizer_prepare(ws, R, idx=idx, field_off=foff, max_field_rows=mfr))
    t = [1]
    def step():
        ops.sparse_optimizer_step(idx, foff, d_rows, rows, mom, ws, t[0], "adam", 0.01, g_bias=gb, g_lin=gb,
                                  max_field_rows=mfr)
        t[0] += 1
    ts = timed(step)
    print(f"max_field_rows={mfr}: prepare {tp:.1f} us, step {ts:.1f} us", flush=True)
