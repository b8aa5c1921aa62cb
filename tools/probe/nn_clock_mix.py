"""In-kernel clock of dense_nn_kernel (RM_NN_STAMP build) when it runs inside a longer back-to-back mix:
MIX=nn (only the NN call), gemm (the six GEMM calls of DCN's MLP), step (the whole DCN fwd+bwd, eager)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recman_amd import _lib, ops  # noqa: E402

mix, reps = os.environ.get("MIX", "gemm"), int(os.environ.get("REPS", 60))
B, FD, Dn, H = 131072, 416, 13, 400
dev = "cuda"
xe, xd = torch.randn(B, FD, device=dev), torch.randn(B, Dn, device=dev)
W1, W2 = torch.randn(FD + Dn, H, device=dev) * 0.05, torch.randn(H, H, device=dev) * 0.05
b1 = torch.zeros(H, device=dev)
h1, h2, dh = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev), torch.randn(B, H, device=dev)
dxe = torch.empty(B, FD, device=dev)
dW1, dW2 = torch.empty(FD + Dn, H, device=dev), torch.empty(H, H, device=dev)
fws = torch.empty(ops.dense_filter_workspace(FD + Dn, 448), device=dev)
wws = torch.empty(max(ops.dense_wgrad_workspace(FD + Dn, H, B), ops.dense_wgrad_workspace(H, H, B)), device=dev)
nn = lambda: ops.dense_fwd(h1, None, W2, h2, fws, bias=b1, act="relu")
def gemms():
    ops.dense_fwd(xe, xd, W1, h1, fws, bias=b1, act="relu")
    ops.dense_fwd(dh, None, W2, h2, fws, transposed=True, epilogue=ops.DENSE_MUL_ACTGRAD, act="relu", aux1=h1)
    ops.dense_fwd(dh, None, W1[:FD], dxe, fws, transposed=True, epilogue=ops.DENSE_ADD)
    ops.dense_wgrad(xe, xd, dh, dW1, wws)
    ops.dense_wgrad(h1, None, dh, dW2, wws)
    nn()
if mix == "step":
    from bench import synth_inputs, make_engine  # noqa
for _ in range(reps):
    nn() if mix == "nn" else gemms()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 2048
buf = (ctypes.c_ulonglong * (8 * n))()
assert lib.rm_debug_nn_stamps(buf, 8 * n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
ghz = s[:, 7] / np.maximum(s[:, 2] - s[:, 1], 1) * 0.1
print(f"mix {mix} x {reps}: clock in the NN chunk loops {np.median(ghz):.3f} GHz; kernel span {(s[:, 3].max() - s[:, 0].min()) / 100:.1f} us")
