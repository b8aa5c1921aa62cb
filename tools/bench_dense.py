"""Micro-benchmark of the wide dense-layer kernels at DCN's shapes.  Prints the steady-state device time of
every call (median of the last `reps` launches, HIP events on the launch stream; the first launches run on a
cold clock and are 10-15 % slower).  Under rocprofv3 --kernel-trace --stats the kernel durations are the
measurement; `--mm` adds the hipBLASLt calls for comparison."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recman_amd import ops  # noqa: E402

B, FD, Dn, H = int(os.environ.get("DENSE_B", 131072)), 416, 13, 400
dev = "cuda"
xe, xd = torch.randn(B, FD, device=dev), torch.randn(B, Dn, device=dev)
W1, W2 = torch.randn(FD + Dn, H, device=dev) * 0.05, torch.randn(H, H, device=dev) * 0.05
b1 = torch.zeros(H, device=dev)
h1, h2, dh = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev), torch.randn(B, H, device=dev)
dxe = torch.empty(B, FD, device=dev)
dW1, dW2 = torch.empty(FD + Dn, H, device=dev), torch.empty(H, H, device=dev)
fws = torch.empty(ops.dense_filter_workspace(FD + Dn, 448), device=dev)
wws = torch.empty(max(ops.dense_wgrad_workspace(FD + Dn, H, B), ops.dense_wgrad_workspace(H, H, B)), device=dev)
x = torch.cat([xe, xd], 1)
calls = [
    ("nn  fwd0  [B,429]x[429,400] bias relu", 2.0 * B * 429 * 400,
     lambda: ops.dense_fwd(xe, xd, W1, h1, fws, bias=b1, act="relu")),
    ("nn  fwd1  [B,400]x[400,400] bias relu", 2.0 * B * 400 * 400,
     lambda: ops.dense_fwd(h1, None, W2, h2, fws, bias=b1, act="relu")),
    ("nn  bwd1  [B,400]x[400,400]^T * act'", 2.0 * B * 400 * 400,
     lambda: ops.dense_fwd(dh, None, W2, h2, fws, transposed=True, epilogue=ops.DENSE_MUL_ACTGRAD, act="relu", aux1=h1)),
    ("nn  dX    [B,400]x[416,400]^T", 2.0 * B * 400 * 416,
     lambda: ops.dense_fwd(dh, None, W1[:FD], dxe, fws, transposed=True, epilogue=ops.DENSE_ADD)),
    ("tn  dW1   [B,429]^T x [B,400]", 2.0 * B * 429 * 400, lambda: ops.dense_wgrad(xe, xd, dh, dW1, wws)),
    ("tn  dW2   [B,400]^T x [B,400]", 2.0 * B * 400 * 400, lambda: ops.dense_wgrad(h1, None, dh, dW2, wws)),
]
if "--mm" in sys.argv:
    calls += [("mm  x @ W1 (hipBLASLt)", 2.0 * B * 429 * 400, lambda: torch.mm(x, W1, out=h1)),
              ("mm  x^T @ dh (hipBLASLt)", 2.0 * B * 429 * 400, lambda: torch.mm(x.t(), dh, out=dW1))]
warm, reps = 12, 12
for _ in range(warm):
    for _, _, fn in calls:
        fn()
torch.cuda.synchronize()
total = 0.0
for name, flop, fn in calls:
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    total += med
    print(f"{name:42s} {med:8.1f} us  {flop / med / 1e6:6.1f} TFLOP/s  {flop / med / 1e6 / 157.3:5.3f} of the f32 MFMA peak")
print(f"sum {total:.1f} us   lib {os.environ.get('RECMAN_HIP_LIB', 'product')}")
# the same six calls back to back without a host synchronisation in between (the state a training loop is in:
# the chip raises its clock only under sustained load)
for rounds in (5, 20, 100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        for _, _, fn in calls[:6]:
            fn()
    e1.record()
    e1.synchronize()
    print(f"sustained: {rounds:3d} rounds of the six calls, {e0.elapsed_time(e1) * 1e3 / rounds:8.1f} us per round")
