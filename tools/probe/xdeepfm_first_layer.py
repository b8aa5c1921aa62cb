"""xDeepFM configs[2] step with the first CIN layer on the symmetric f32 kernels or on the split-operand ones (default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import engine as eng

B, F, V, Dn, D = 65536, 26, 1_000_001, 13, 16
g = torch.Generator(device="cuda").manual_seed(1)
idx = torch.randint(0, V, (B, F), generator=g, device="cuda")
dense = torch.randn(B, Dn, generator=g, device="cuda")
y = (torch.rand(B, generator=g, device="cuda") < 0.25).long()
for first in ("sym", "bf16x6"):
    spec = eng.FeatureSpec([f"C{i}" for i in range(F)], [V] * F, [f"I{j}" for j in range(Dn)])
    e = eng.XDeepFMEngine(spec, D, dict(deep_hidden_units=(32, 32), deep_activation="leaky_relu",
                                        cin_cross_layer_units=[128, 128], cin_activation="leaky_relu",
                                        cin_first_layer=first))
    e.rows.normal_(0, 0.01, generator=g)
    for k, p in e.params.items():
        if k.startswith("cin_filter") or k.startswith("dnn_layer") and k.endswith("weights"):
            p.normal_(0, 0.05, generator=g)
    for _ in range(3):
        loss = e.fwd_bwd(idx, dense, y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        loss = e.fwd_bwd(idx, dense, y)
    b.record()
    torch.cuda.synchronize()
    print(f"cin_first_layer={first}: {a.elapsed_time(b) / 10:.3f} ms per step, loss {float(loss):.6f}", flush=True)
    del e
    torch.cuda.empty_cache()
