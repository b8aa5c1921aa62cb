"""Can a torch.distributed (RCCL) all_to_all_single be captured in a hipGraph?  Run with
RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 python tools/probe/rccl_graph_probe.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29512")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.arange(1024, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
dist.all_to_all_single(y, x)  # warm-up (communicator creation is not capturable)
dist.all_reduce(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            z = x * 2
            dist.all_to_all_single(y, z)
            w = y + 1
    torch.cuda.synchronize()
    x.fill_(3.0)
    g.replay()
    torch.cuda.synchronize()
    print("capture ok; replay value", float(w[0]), "(expect 7.0)")
except Exception as e:  # noqa: BLE001
    print("capture FAILED:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
