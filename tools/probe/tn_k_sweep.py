import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import ops
B = 131072
dev = "cuda"
def timeit(fn, n=8):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / n * 1e3
for K in (384, 400, 416, 512):
    A = torch.randn(B, K, device=dev); G = torch.randn(B, 400, device=dev)
    dW = torch.empty(K, 400, device=dev)
    ws = torch.empty(ops.dense_wgrad_workspace(K, 400, B), device=dev)
    t = timeit(lambda: ops.dense_wgrad(A, None, G, dW, ws))
    print(f"TN K={K} N=400: {t:7.1f} us  {2.0*B*K*400/t/1e6:6.1f} TFLOP/s")
