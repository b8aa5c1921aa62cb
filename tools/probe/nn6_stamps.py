"""Shader clock and block duration of the bf16x6 NN kernel (a -DRM_NN6_STAMP build, RECMAN_HIP_LIB)."""
import ctypes, os, sys
os.environ["NN6_ONE"] = "1"
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import numpy as np
import nn6_time as T   # runs the one shape
n = 4 * 1024
buf = (ctypes.c_ulonglong * n)()
T.L.rm_debug_nn6_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert T.L.rm_debug_nn6_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.float64)
rt, st = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
ok = rt > 0
print(f"blocks {ok.sum()}: block duration {np.median(rt[ok]) / 100:.1f} us (median), shader clock "
      f"{np.median(st[ok] / rt[ok]) * 100:.0f} MHz (median), min {np.min(st[ok] / rt[ok]) * 100:.0f}, max {np.max(st[ok] / rt[ok]) * 100:.0f}")
t0 = a[ok, 0].min()
print("first / last block start (us):", (a[ok, 0].min() - t0) / 100, (a[ok, 0].max() - t0) / 100, "last end", (a[ok, 2].max() - t0) / 100)
