"""Eager time of the one-kernel DeepFM step (configs[1]) for the library named by RECMAN_HIP_LIB: kernel-only timing
with HIP events over 200 launches after a warm-up (ablation builds give wrong results; this only times them)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

a = bench.parse(["--workload", "deepfm", "--only", "--no-pmc", "--no-optimizer", "--no-cpu-baseline"] + sys.argv[1:])
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["deepfm"]
w = dict(w, hp=dict(w["hp"], step_fusion=True))
engine, idx, dense, y, hp = bench.make_engine(a, w, w["B"], w["V"], dev, 0, 1, False, zipf=a.zipf)
for _ in range(300):
    engine.fwd_bwd(idx, dense, y)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(200):
    engine.fwd_bwd(idx, dense, y)
ev[1].record()
torch.cuda.synchronize()
print(f"{os.environ.get('RECMAN_HIP_LIB', 'product')}: step {ev[0].elapsed_time(ev[1]) / 200 * 1e3:.1f} us")
