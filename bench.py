#!/usr/bin/env python
"""Headline benchmark: examples/sec of one forward+backward pass (loss -> gradient
of every parameter, optimizer step excluded) on synthetic Criteo-shaped batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload deepfm|xdeepfm|dcn]

N = 1 runs in this process.  N > 1: one rank per GPU over RCCL, the embedding table
row-sharded over the ranks (recman_amd/dist.py), per-GPU batch fixed (weak scaling).
The ranks are started either by the driver (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`: WORLD_SIZE is set) or, when WORLD_SIZE is
unset, by this script itself: BEFORE any GPU call it starts `torch.distributed.run`
as a child process, relays rank 0's JSON line and exits with the child's code.
Rank 0 prints ONE JSON line.

The default run (N = 1, no --workload) measures BASELINE configs[1] (DeepFM) as the
contract line - `value`, `roofline`, `cpu_baseline` - and adds, in the same line,
`workloads.xdeepfm` (configs[2], CIN on the f32 MFMA roofline) and `workloads.dcn`
(configs[3], dense GEMM on MFMA + the fused cross kernels on HBM), each with its own
`ms_per_step`, `value`, `roofline(s)` and `cpu_baseline`, plus `zipf` (the DeepFM step
with Zipf(1.05) ids).  `--workload X` (or --only) measures that workload alone.

`roofline.traffic`: HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and
WRITE_SIZE each in its own `--pmc ... --kernel-trace` run, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950) of THIS build, collected live by two child
processes (`bench.py --pmc-child`, the roofline kernels only) started before this
process touches the GPU; `traffic_live` says whether that worked - otherwise the value
comes from the committed profile named in `traffic_source`, or is null.

Workloads (BASELINE.json `configs`), seed 2019, dropout keep 1,
embedding_l2_reg 0, weights N(0, 0.01), indices uniform over each field's vocabulary:
  deepfm  (default) configs[1]: 26 sparse x 1,000,001 rows + 13 dense, D=16, MLP (32,32) relu, B=65536
  xdeepfm           configs[2]: same inputs, CIN [128,128] leaky_relu, MLP (32,32) leaky_relu
  dcn               configs[3]: 6 vector cross layers + MLP [400,400] relu, B=131072
  xdeepfm_100m      configs[4]: xDeepFM on a 100 M-row x 64-dim table (row-sharded with --gpus 8)
  dcn_matrix        (extra)    : the same with matrix cross layers x0 o (W x_l + b) + x_l
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3
METRIC = "examples/sec fwd+bwd, Criteo-shape batch 65536; % HBM and MFMA roofline"

WORKLOADS = {
    "deepfm": dict(model="deepfm", B=65536, D=16, F=26, V=1_000_001, Dn=13,
                   hp=dict(deep_hidden_units=(32, 32), deep_activation="relu")),
    "xdeepfm": dict(model="xdeepfm", B=65536, D=16, F=26, V=1_000_001, Dn=13,
                    hp=dict(deep_hidden_units=(32, 32), deep_activation="leaky_relu",
                            cin_cross_layer_units=(128, 128), cin_activation="leaky_relu")),
    "dcn": dict(model="dcn", B=131072, D=16, F=26, V=1_000_001, Dn=13,
                hp=dict(deep_hidden_units=(400, 400), deep_activation="relu", cross_layer_num=6)),
    # BASELINE configs[4]: 100 M rows x 64 floats, meant for --gpus 8 (row-sharded; 3.4 GB per shard);
    # also runs on one GPU (51 GB of fused rows)
    "xdeepfm_100m": dict(model="xdeepfm", B=65536, D=64, F=26, V=3_846_154, Dn=13,
                         hp=dict(deep_hidden_units=(32, 32), deep_activation="leaky_relu",
                                 cin_cross_layer_units=(128, 128), cin_activation="leaky_relu")),
    # not a BASELINE config: the matrix form of the cross layers (x0 o (W x_l + b) + x_l), same shapes
    "dcn_matrix": dict(model="dcn", B=131072, D=16, F=26, V=1_000_001, Dn=13,
                       hp=dict(deep_hidden_units=(400, 400), deep_activation="relu", cross_layer_num=6,
                               cross_type="matrix")),
}
CONFIG_NO = {"deepfm": 1, "xdeepfm": 2, "dcn": 3, "xdeepfm_100m": 4}
# committed PMC profiles of earlier builds: the fallback when the live PMC passes are unavailable
TRAFFIC_FALLBACK = os.path.join(ROOT, "profiles", "traffic.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--table-row-reuse", default="", choices=["", "stream", "cache"],
                    help="gather kernel's row loads: stream = non-temporal (default for uniform ids), "
                         "cache = plain (default with --zipf)")
    ap.add_argument("--d-rows-reuse", default="cache", choices=["cache", "stream"],
                    help="the backward's row-gradient stores: cache (default - what fit() runs: the optimizer step "
                         "gathers them right after) or stream (non-temporal: a bare fwd+bwd is ~1 %% faster)")
    ap.add_argument("--prewarm", type=float, default=0.25,
                    help="seconds of untimed steps BEFORE the W warmup steps (GPU clock ramp); 0 = none")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="measure this workload alone (default: deepfm as the contract line + the "
                         "xdeepfm / dcn / zipf sub-records)")
    ap.add_argument("--only", action="store_true",
                    help="no sub-records: the named (or default deepfm) workload alone")
    ap.add_argument("--rotate", type=int, default=8,
                    help="distinct synthetic batches the timed steps cycle through (single GPU; fit() never sees the "
                         "same ids twice in a row: a replayed batch could find its table rows in the 256 MiB Infinity "
                         "Cache); 1 = replay one batch")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (tests)")
    ap.add_argument("--vocab", type=int, default=0, help="override rows per field (tests)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="skip the separately-reported optimizer step")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not start the rocprofv3 --pmc child passes (roofline.traffic then comes from "
                         "the committed profile, tagged traffic_live=false)")
    ap.add_argument("--pmc-child", default="", help=argparse.SUPPRESS)  # comma list of workloads
    # (tests) the rank runtime alone, on CPU with gloo: every rank all_reduces in a loop; "die:R:K" makes rank R
    # exit abruptly before iteration K, "raise:R:K" makes it raise there
    ap.add_argument("--selftest-ranks", default="", help=argparse.SUPPRESS)
    ap.add_argument("--zipf", type=float, default=0.0, help="Zipf exponent for indices (0 = uniform)")
    ap.add_argument("--cpu-budget", type=float, default=0.0,
                    help="seconds of CPU-baseline work per workload (default 20 for the contract line, "
                         "12 for the sub-records)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the row-sharded engine even at world size 1 (rehearsal)")
    ap.add_argument("--no-graph-segments", action="store_true",
                    help="sharded + fixed-capacity: do not capture the compute between the collectives "
                         "as hipGraph segments (eager kernel launches instead)")
    ap.add_argument("--graph-sharded", action="store_true",
                    help="capture the WHOLE row-sharded step (fixed-capacity exchange, RCCL calls included) "
                         "in one hipGraph; opt-in - RCCL capture was only exercised at world size 1 "
                         "(the default captures the compute between the collectives instead)")
    ap.add_argument("--micro-batches", type=int, default=0,
                    help="row-sharded engine: pipeline the step over this many micro-batches so the "
                         "all_to_alls overlap the compute (0 = auto: 2 when N > 1, else 1)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "fixed", "dynamic"],
                    help="row-sharded exchange layout: fixed capacity (equal splits, no host sync, one "
                         "hipGraph per step; falls back when a batch overflows) or dynamic split sizes; "
                         "auto = fixed for uniform indices, dynamic for --zipf")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------- self-launch (N > 1)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n, argv, port=None, python=None):
    """The torch.distributed.run command line that starts n ranks of this script."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()),
            os.path.abspath(__file__), *argv]


def self_launch(n, argv):
    """--gpus N > 1 without WORLD_SIZE: start the N ranks as CHILD processes (never exec, and before
    this process has made any GPU call), relay rank 0's JSON line, return the child's exit code."""
    cmd = launch_command(n, argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for out in p.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("[bench] the ranks exited 0 without a JSON line", file=sys.stderr)
        rc = 1
    return rc


# ---------------------------------------------------------------------- inputs
def synth_inputs(w, B, V, device, seed, zipf=0.0):
    g = torch.Generator(device=device).manual_seed(seed)
    if zipf > 0:
        # inverse-CDF sampling of a truncated Zipf(s) over [1, V]; slot 0 stays the null row
        u = torch.rand(B, w["F"], generator=g, device=device, dtype=torch.float64)
        s = zipf
        idx = ((u * (float(V) ** (1 - s) - 1) + 1) ** (1 / (1 - s))).long().clamp_(1, V - 1)
    else:
        idx = torch.randint(0, V, (B, w["F"]), generator=g, device=device, dtype=torch.int64)
    dense = torch.randn(B, w["Dn"], generator=g, device=device, dtype=torch.float32)
    y = (torch.rand(B, generator=g, device=device) < 0.25).long()
    return idx, dense, y


def init_engine(engine, seed):
    """weights ~ N(0, 0.01) written in place (tables in chunks: 3.3 GB of fused rows at config 2)."""
    g = torch.Generator(device=engine.device).manual_seed(seed)
    for base in engine.storage():
        flat = base.view(-1)
        for s in range(0, flat.numel(), 1 << 26):
            e = min(flat.numel(), s + (1 << 26))
            flat[s:e] = torch.randn(e - s, generator=g, device=engine.device) * 0.01


def cpu_baseline(w, hp, idx, dense, y, engine, budget_s=20.0):
    """The CPU PyTorch restatement (oracle/, kind "port") on the host cores, same
    inputs and weights, fwd+bwd, sparse embedding gradients (what TF's IndexedSlices
    are with embedding_l2_reg = 0).  BOUNDED: a probe pass on 1024 examples sizes the
    sample so that the timed passes take about `budget_s` seconds in total."""
    from oracle import th_layers as T

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    spec = T.Spec(engine.spec.sparse_names, engine.spec.feat_sizes, engine.spec.dense_names)
    p = {k: v.detach().cpu() for k, v in engine.state_dict().items()}
    B = idx.shape[0]
    idx_c, dense_c, y_c = idx.cpu(), dense.cpu(), y.cpu()
    chunk_max = 4096 if w["model"] == "xdeepfm" else 1 << 30

    def step(sample):
        leaves = {k: v.requires_grad_(True) for k, v in p.items()}
        for v in leaves.values():
            v.grad = None
        logits = []
        chunk = min(sample, chunk_max)
        for s0 in range(0, sample, chunk):
            sl = slice(s0, min(sample, s0 + chunk))
            loss, logit, _ = T.model_loss(w["model"], leaves, spec, idx_c[sl], dense_c[sl], y_c[sl],
                                          hp, sparse_grad=True)
            (loss * (sl.stop - sl.start) / sample).backward()
            logits.append(logit.detach().reshape(-1))
        return torch.cat(logits)

    step(min(B, 1024))
    t0 = time.perf_counter()
    step(min(B, 1024))
    probe = time.perf_counter() - t0
    iters = 3
    sample = int(min(B, max(1024, 1024 * (budget_s / (iters + 1)) / max(probe, 1e-4))))
    sample = max(1024, (sample // 1024) * 1024) if B >= 1024 else B
    t0 = time.perf_counter()
    step(sample)
    t1 = time.perf_counter() - t0
    # the 1024-example probe is overhead-dominated and undersizes the sample: resize once from a pass
    # at the first size so that the timed passes really take about budget_s
    if B >= 1024:
        want = int(min(B, sample * (budget_s / (iters + 1)) / max(t1, 1e-4)) // 1024 * 1024)
        if want > 1.5 * sample:
            sample = want
            step(sample)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        logit = step(sample)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return dict(value=round(sample / med, 1), unit="examples/s", cores=torch.get_num_threads(),
                host_cpu_count=os.cpu_count(), torch_threads=torch.get_num_threads(),
                kind="port",
                sample=f"first {sample} examples of the same batch, {iters} timed fwd+bwd passes "
                       f"(median {med:.2f} s; SURVEY.md 8d asks for >= 10 passes on the whole batch - bounded "
                       f"here to ~{budget_s:.0f} s of CPU work so that the default run finishes in minutes), "
                       f"oracle/th_layers.py on torch CPU with {torch.get_num_threads()} threads of the box's "
                       f"{os.cpu_count()} logical CPUs, sparse embedding grads"
                       + (f", CIN in chunks of {chunk_max}" if chunk_max < sample else "")), logit, sample


# ---------------------------------------------------------------------- PMC traffic (child passes)
PMC_LAUNCHES = 3  # launches of each roofline kernel a --pmc-child makes at the end of its run


def under_profiler():
    env = os.environ
    return ("rocprof" in env.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in env
            or "ROCPROFILER_REGISTER_FORCE_LOAD" in env or "ROCPROF_OUTPUT_PATH" in env)


def pmc_passes(workloads, timeout_s=240):
    """Runs `rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py --pmc-child <workloads>` once
    per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass) and returns
    {workload: {kernel symbol: {"FETCH_SIZE": kb, "WRITE_SIZE": kb}}} averaged over the PMC_LAUNCHES
    launches the child makes of each roofline kernel between two marker kernels; {} (+ a reason) when
    rocprofv3 is unavailable or fails.
    Must run BEFORE this process makes a GPU call (the children use the GPU alone)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or under_profiler():
        return {}, "rocprofv3 not available (or this run is itself profiled)"
    out = {}
    tmp = tempfile.mkdtemp(prefix="recman_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            plan = os.path.join(tmp, f"plan_{counter}.json")
            # the program after `--` is THIS interpreter's real binary: a python3 found on PATH may be a wrapper
            # script (an exec behind the profiler's preloaded library) or another torch build
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   os.path.realpath(sys.executable), os.path.abspath(__file__),
                   "--pmc-child", ",".join(workloads), "--no-pmc"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(env, RECMAN_PMC_PLAN=plan), timeout=timeout_s,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0 or not os.path.exists(plan):
                return {}, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {r.stderr[-300:]}"
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                rows += list(csv.DictReader(open(f)))
            err = reduce_counter(rows, json.load(open(plan)), counter, out)
            if err:
                return {}, err
        return out, None
    except Exception as e:  # a diagnostic extra: never let it break the contract line
        return {}, f"PMC passes failed: {type(e).__name__}: {str(e)[:200]}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


MARKER = "rm_profile_marker_kernel"


def reduce_counter(rows, plan, counter, out):
    """rows: the dicts of a rocprofv3 counter_collection.csv; plan: [(workload, kernel symbol), ...] in
    the order the child ran its probes, each bracketed by two rm_profile_marker launches.  Stores the
    average Counter_Value of the probe's launches in out[workload][symbol][counter].  Returns an
    error text or None."""
    rows = sorted((x for x in rows if x.get("Counter_Name") == counter), key=lambda x: int(x.get("Dispatch_Id", 0)))
    marks = [i for i, x in enumerate(rows) if MARKER in x["Kernel_Name"]]
    if len(marks) != 2 * len(plan):
        return f"PMC pass ({counter}): {len(marks)} marker dispatches for {len(plan)} probes"
    for k, (wname, symbol) in enumerate(plan):
        seg = rows[marks[2 * k] + 1: marks[2 * k + 1]]
        vals = [float(x["Counter_Value"]) for x in seg if symbol in x["Kernel_Name"]]
        if len(vals) != PMC_LAUNCHES:
            return f"PMC pass ({counter}): {len(vals)} launches of {symbol} between its markers, expected {PMC_LAUNCHES}"
        out.setdefault(wname, {}).setdefault(symbol, {})[counter] = sum(vals) / len(vals)
    return None


def pmc_child(a):
    """The profiled child: per workload one engine, one warm step, then PMC_LAUNCHES launches of each
    roofline kernel (the parent reads the counters of exactly those dispatches)."""
    from recman_amd import ops

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    plan = []
    for name in a.pmc_child.split(","):
        w = WORKLOADS[name]
        engine, idx, dense, y, _ = make_engine(a, w, w["B"], w["V"], dev, 0, 1, False)
        engine.fwd_bwd(idx, dense, y)
        torch.cuda.synchronize()
        for p in engine.roofline_probes(idx, dense, y):
            ops.profile_marker(len(plan))
            for _ in range(PMC_LAUNCHES):
                p["fn"]()
            ops.profile_marker(len(plan))
            torch.cuda.synchronize()
            plan.append((name, p["symbol"]))
        del engine, idx, dense, y
        torch.cuda.empty_cache()
    with open(os.environ["RECMAN_PMC_PLAN"], "w") as f:
        json.dump(plan, f)


def attach_traffic(roofs, wname, live, fallback):
    """Adds traffic (HBM bytes per launch: FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes) to each roofline."""
    for r in roofs:
        c = (live.get(wname) or {}).get(r["symbol"])
        if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            r["traffic"] = round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, 0)
            r["traffic_live"] = True
            r["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this run "
                                   f"(avg of {PMC_LAUNCHES} launches; FETCH_SIZE x2 per MI355X_MICROARCH.md)")
        else:
            fb = (fallback.get(wname) or {}).get(r["symbol"])
            r["traffic"] = fb["bytes"] if fb else None
            r["traffic_live"] = False
            if fb:
                r["traffic_source"] = fb["source"]
        if r["traffic"] and r["bound"] == "hbm":
            # the same launch priced on the bytes the counters saw instead of the algorithmic ones
            r["frac_traffic"] = round(r["traffic"] / (r["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)


# ---------------------------------------------------------------------- one workload
def make_engine(a, w, B, V, dev, rank, world, sharded, zipf=0.0):
    from recman_amd import engine as eng

    spec = eng.FeatureSpec([f"C{i + 1}" for i in range(w["F"])], [V] * w["F"],
                           [f"I{j + 1}" for j in range(w["Dn"])])
    hp = dict(w["hp"], embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_l2_reg=0.0, cin_l2_reg=0.0,
              cross_layer_l2_reg=0.0,
              # uniform ids: a row is touched about once per batch -> streamed (non-temporal) row loads;
              # --zipf: hot rows want the caches
              table_row_reuse=a.table_row_reuse or ("cache" if zipf > 0 else "stream"),
              d_rows_reuse=a.d_rows_reuse)
    if sharded:
        from recman_amd import dist as rdist

        fixed = a.exchange == "fixed" or (a.exchange == "auto" and zipf == 0)
        micro = a.micro_batches or (2 if world > 1 else 1)
        engine = rdist.make_sharded_engine(w["model"], spec, w["D"], hp, dev, rank, world,
                                           capacity_factor=1.0 if fixed else None, micro_batches=micro)
    else:
        engine = eng.ENGINES[w["model"]](spec, w["D"], hp, device=dev)
    init_engine(engine, 2019)
    idx, dense, y = synth_inputs(w, B, V, dev, 2019 + rank, zipf)
    return engine, idx, dense, y, hp


def time_steps(a, engine, idx, dense, y, sharded, world, dist, rehearse, dev, steps, warmup, batches=None):
    """W warmup + K timed steps of engine.fwd_bwd -> (ms per step, host enqueue ms, graph kind, fixed, rotation).
    batches: further (idx, dense, y) triples - the timed steps then cycle through all of them (one captured
    hipGraph per batch, replayed round robin), and the single-batch replay is timed beside it."""
    # (the captured graphs / replay closures are dropped on return: they pin the capture's memory pool)
    allb = [(idx, dense, y)] + list(batches or [])

    def step(k=0):
        return engine.fwd_bwd(*allb[k])

    # ---- optional hipGraph capture of the whole step (launch-bound otherwise) ----
    # (the dynamic exchange needs host-side split sizes and cannot be captured)
    fixed = sharded and engine.st.capacity_factor is not None
    use_graph = not a.no_graph and (not sharded or (fixed and a.graph_sharded))
    graphs = None
    step()
    torch.cuda.synchronize()

    def any_overflow():
        """Fixed-capacity exchange only: did a batch on ANY rank exceed its bucket capacity?"""
        if not fixed:
            return False
        hit = torch.tensor([1.0 if engine.overflowed() else 0.0], device=dev)
        if dist is not None and world > 1:
            hit = hit.cpu() if rehearse else hit
            dist.all_reduce(hit, op=dist.ReduceOp.MAX)
        return bool(hit.item() > 0)

    if any_overflow():  # skewed indices: the fixed layout does not fit, use exact split sizes
        print("[bench] fixed-capacity exchange overflowed, using dynamic split sizes", file=sys.stderr)
        engine.st.capacity_factor, engine._B, fixed, use_graph = None, None, False, False
        step()
        torch.cuda.synchronize()
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            import gc

            graphs = []
            gc.collect()
            gc.disable()  # (a collection inside the capture could free device memory: unsafe there)
            try:
                for k in range(len(allb)):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        step(k)
                    graphs.append(g)
            finally:
                gc.enable()
        except Exception as e:  # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture failed, running eager: {e}", file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()
    segments = False
    if sharded and fixed and graphs is None and not a.no_graph and not a.no_graph_segments:
        # the compute between the collectives as hipGraphs, the RCCL calls eager in between
        try:
            engine.capture_segments(idx, dense, y)
            segments = True
        except Exception as e:
            print(f"[bench] segment capture failed, running eager: {e}", file=sys.stderr)
            engine._segs = None
            torch.cuda.synchronize()
    nrot = 1 if segments else len(allb)  # (captured segments are bound to the first batch's buffers)

    def run(i=0):
        if graphs is not None:
            graphs[i % nrot].replay()
        else:
            step(i % nrot)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: a step takes 0.1 ms, so W warmup steps alone end before the GPU has left its idle
    # clocks (50 timed steps measured 2.5 % slower than 1000).  Untimed, before the W warmup steps.
    if a.prewarm > 0 and sharded:
        # (a FIXED count with collectives inside: every rank must issue the same number of them)
        for i in range(100):
            run(i)
        torch.cuda.synchronize()
    elif a.prewarm > 0:
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < a.prewarm:
            for i in range(24):
                run(i)
            torch.cuda.synchronize()

    def timed(rotating):
        for i in range(warmup):
            run(i if rotating else 0)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            run(i if rotating else 0)
        enq = time.perf_counter() - t0  # host time to enqueue the steps (launch-bound when ~= elapsed)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device="cpu" if rehearse else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
        return el / steps * 1e3, enq / steps * 1e3

    rotation = None
    if nrot > 1:
        ms_one, _ = timed(False)
        ms, enq_ms = timed(True)
        rotation = {"batches": nrot, "ms_per_step_one_batch_replayed": round(ms_one, 4),
                    "ms_per_step_rotating": round(ms, 4),
                    "what": f"the timed steps cycle through {nrot} distinct synthetic batches (one captured hipGraph "
                            "each, replayed round robin): `value` is the ROTATING number; the single-batch replay "
                            "beside it shows what a resident batch would be worth"}
    else:
        ms, enq_ms = timed(False)
    if any_overflow():
        raise SystemExit("[bench] a timed batch overflowed the fixed-capacity exchange: rerun with --exchange dynamic")
    kind = "segments between the collectives" if segments else graphs is not None
    return ms, enq_ms, kind, fixed, rotation


def train_step_record(a, w, engine, dev, B, V, steps=60):
    """What fit() achieves (recman/tf/core/DeepModel.py:180-202, xDeepFM.py:116-126): the loop
    DeepModel._fit_encoded runs - a NEW batch every step -> forward+backward -> row-wise optimizer step on the
    touched table rows (its id-only sort issued on a side stream beside fwd+bwd) -> dense parameters - over
    `steps` steps, with the encoded dataset (a) resident in HBM and (b) in pinned host memory behind
    th/feeder.py's double-buffered H2D copies.  Reported, NOT part of `value`."""
    from recman_amd.optim import FusedDenseOptimizer, SparseTableOptimizer
    from recman_amd.th.feeder import BatchFeeder

    nb = 16  # distinct batches of the synthetic dataset
    g = torch.Generator().manual_seed(4242)
    idx_h = torch.randint(0, V, (nb * B, w["F"]), generator=g, dtype=torch.int64)
    dense_h = torch.randn(nb * B, w["Dn"], generator=g)
    y_h = (torch.rand(nb * B, generator=g) < 0.25).long()
    sopt, dopt = SparseTableOptimizer(engine, "adam", 1e-3), FusedDenseOptimizer(engine, "adam", 1e-3)
    side = torch.cuda.Stream(device=dev)
    sopt._workspace(B * w["F"])

    def one(ib, db, yb):
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            sopt.prepare(ib)
        engine.fwd_bwd(ib, db, yb)
        torch.cuda.current_stream(dev).wait_stream(side)
        sopt.step(ib)
        dopt.step()

    def timed(batches, n):
        it = iter(batches())
        for _ in range(4):
            one(*next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        for ib, db, yb in it:
            one(ib, db, yb)
            done += 1
            if done == n:
                break
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / max(done, 1) * 1e3, done

    # (a) the dataset resident in HBM
    dev_b = [(idx_h[k * B:(k + 1) * B].to(dev), dense_h[k * B:(k + 1) * B].to(dev), y_h[k * B:(k + 1) * B].to(dev))
             for k in range(nb)]

    def resident():
        while True:
            yield from dev_b

    ms_res, n_res = timed(resident, steps)
    del dev_b
    # (b) pinned host memory -> H2D on a copy stream, one batch ahead
    feeder = BatchFeeder(idx_h, dense_h, y_h, B, dev)

    def pinned():  # (a shuffled epoch, as fit() runs it: the rows of a batch are gathered on the host)
        while True:
            for _, _, ib, db, yb in feeder.batches(perm=torch.randperm(nb * B, generator=g)):
                yield ib, db, yb

    ms_pin, n_pin = timed(pinned, steps)
    per_batch = B * (w["F"] * 8 + w["Dn"] * 4 + 8)
    return {"ms_per_step_dataset_in_hbm": round(ms_res, 4), "examples_per_s_dataset_in_hbm": round(B / ms_res * 1e3, 1),
            "ms_per_step_pinned_feeder": round(ms_pin, 4), "examples_per_s_pinned_feeder": round(B / ms_pin * 1e3, 1),
            "steps": n_res, "distinct_batches": nb, "host_to_device_bytes_per_step": per_batch,
            "pcie_floor_ms_at_63GBs": round(per_batch / 63e9 * 1e3, 4),
            "what": "fit()'s loop at this workload: new batch -> fwd+bwd -> row-wise lazy Adam on the touched rows "
                    "(sort on a side stream beside fwd+bwd) -> dense Adam; eager launches, every step on other "
                    "ids.  With the dataset in pinned host memory the step is bound by the H2D copy of the ids "
                    "(int64, the reference's dtype), not by the GPU: compare pcie_floor_ms.  NOT part of value"}


def step_hbm(w, B, ms):
    """algorithmic bytes of the embedding+FM forward AND backward per batch
    (idx, rows, E, S, g, dE, gradient rows) over the whole step time."""
    F, D = w["F"], w["D"]
    fwd = B * F * (8 + 4 * D + 4) + B * 4 + B * F * 4 * D
    bwd = B * F * (8 + 4 * D + 4 * D + 4 * D + 4) + B * 4
    gbs = (fwd + bwd) / (ms * 1e-3) / 1e9
    return {"algorithmic_bytes": fwd + bwd, "achieved": round(gbs, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": round(gbs / HBM_PEAK_GBS, 4),
            "what": "embed+FM fwd+bwd algorithmic bytes of SURVEY.md 8d over the WHOLE step's time"}


def measure(a, wname, dev, rank, world, dist, rehearse, sharded, live_traffic, fallback, steps, warmup,
            cpu_budget, zipf=0.0, want_cpu=True, want_opt=True):
    """One workload at the current world size -> its record (dict)."""
    w = dict(WORKLOADS[wname])
    B = a.batch or w["B"]
    V = a.vocab or w["V"]
    engine, idx, dense, y, hp = make_engine(a, w, B, V, dev, rank, world, sharded, zipf)
    # further batches to rotate through (the big-table configs[4] keeps one: its table alone is 51 GB)
    nrot = max(1, a.rotate) if (not sharded and w["D"] * w["V"] * w["F"] < 1 << 31) else 1
    batches = [synth_inputs(w, B, V, dev, 2019 + rank + 1000 * k, zipf) for k in range(1, nrot)]
    ms, enq_ms, graph_kind, fixed, rotation = time_steps(a, engine, idx, dense, y, sharded, world, dist, rehearse,
                                                         dev, steps, warmup, batches)
    value = world * B / (ms * 1e-3)

    # ---- rooflines of the hand-written hot kernels, HIP events on the stream they run on ----
    # (sharded: only for the models whose dominant kernel is the gather; rank 0, local kernel only)
    roofs = []
    if not sharded or (rank == 0 and w["model"] == "deepfm"):
        roofs = engine.roofline_probe_all(idx, dense, y, iters=max(10, min(steps, 50)))
        full = not sharded and B == w["B"] and V == w["V"] and zipf == 0
        attach_traffic(roofs, wname, live_traffic if full else {}, fallback if full else {})
    for r in roofs:
        r.pop("fn", None)

    rec = {
        "value": round(value, 1), "unit": "examples/s", "ms_per_step": round(ms, 4),
        "config": {"workload": (f"{wname} (BASELINE configs[{CONFIG_NO[wname]}])" if wname in CONFIG_NO
                                else f"{wname} (extra workload, not a BASELINE config)"),
                   "batch_per_gpu": B, "batch_all_gpus": B * world, "sparse_fields": w["F"],
                   "rows_per_field": V,
                   "dense_fields": w["Dn"], "emb_dim": w["D"], "hp": {k: v for k, v in w["hp"].items()},
                   "indices": "uniform" if zipf == 0 else f"zipf({zipf})",
                   "table_row_loads": (
                       # DeepFM's one-kernel step (rm_deepfm_step) has its own switch, hp step_row_loads: cached by
                       # default (non-temporal row loads measured slower there, profiles/r03_deepfm_step.md)
                       ("cached" if hp.get("step_row_loads", "cache") == "cache" else "non-temporal")
                       + " (rm_deepfm_step)" if getattr(engine, "_step_ok", None) is True
                       else ("cached" if hp["table_row_reuse"] == "cache"
                             else "non-temporal (ids with little reuse per batch)")),
                   "row_gradient_stores": ("cached (as in fit(): the optimizer step gathers them next)"
                                           if hp["d_rows_reuse"] == "cache" else "non-temporal"),
                   "embedding_l2_reg": 0.0, "dropout_keep": 1.0, "optimizer_step": "excluded",
                   # how the GEMM-shaped kernels of this workload multiply: fp32 in, fp32 out, fp32
                   # accumulate everywhere; the wide dense layers and CIN form every fp32 product from three bf16
                   # pieces per operand on the bf16 matrix pipe (six exact piece products, error at fp32 level:
                   # `parity_check` beside this record is measured on THIS path)
                   "matrix_arithmetic": ({"dcn": "fp32 operands split into 3 bf16 pieces, 6 exact piece products per "
                                                 "k-step on the bf16 MFMA, fp32 accumulate (rm_dense_fwd6 / rm_dense_wgrad6)",
                                          "xdeepfm": "CIN: fp32 operands split into 3 bf16 pieces, 6 exact piece products "
                                                     "per k-step on the bf16 MFMA, fp32 accumulate (csrc/cin6.hip; the "
                                                     "first layer's dW and the skinny DNN: f32 MFMA)"}.get(wname,
                                         "f32 MFMA (v_mfma_f32_16x16x4_f32), fp32 accumulate")),
                   "host_enqueue_ms_per_step": round(enq_ms, 4), "batches_rotated": rotation or 1,
                   "prewarm_s": a.prewarm, "hipgraph": graph_kind,
                   "table": (f"row-sharded mod {world}, fused [D+4] rows, all_to_all over xGMI, "
                             + ("fixed-capacity exchange (equal splits, no host sync)" if fixed
                                else "dynamic split sizes (one host sync per batch)")
                             + f", {engine.micro_batches} micro-batch(es) per step") if sharded
                   else "single GPU"},
        "roofline": roofs[0] if roofs else None,
    }
    if sharded:
        rec["exchange"] = exchange_record(engine, world, (dist.get_backend() if dist is not None else "none"), B, w)
    if len(roofs) > 1:
        rec["rooflines"] = roofs
    if w["model"] == "deepfm" and not sharded:
        rec["step_hbm"] = step_hbm(w, B, ms)
    if rank == 0 and want_opt and not a.no_optimizer:
        # reported separately (the headline metric is fwd+bwd only): the row-wise step on the table
        # rows this batch touched + dense Adam on the dense parameters
        try:
            rec["optimizer_step"] = (engine.optimizer_probe(idx, more_ids=[b[0] for b in batches])
                                     if not sharded else None)
        except Exception as e:  # never let the extra break the contract line
            rec["optimizer_step"] = {"ms": None, "error": f"{type(e).__name__}: {str(e)[:200]}"}
    if (rank == 0 and not sharded and want_opt and not a.no_optimizer and wname == "deepfm" and zipf == 0
            and B == w["B"] and V == w["V"]):
        try:
            rec["train_step"] = train_step_record(a, w, engine, dev, B, V)
        except Exception as e:  # never let the extra break the contract line
            rec["train_step"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    if sharded and want_opt and not a.no_optimizer and hasattr(engine, "optimizer_probe_sharded"):
        # every rank takes part (the dense all_reduce precedes the dense step); rank 0 reports
        try:
            o = engine.optimizer_probe_sharded(idx, dense, y)
            if rank == 0:
                rec["optimizer_step"] = o
        except Exception as e:
            rec["optimizer_step"] = {"ms": None, "error": f"{type(e).__name__}: {str(e)[:200]}"}
    if rank == 0 and not sharded and want_cpu and not a.no_cpu_baseline:
        base, logit_cpu, sample = cpu_baseline(w, hp, idx, dense, y, engine, cpu_budget)
        rec["cpu_baseline"] = base
        engine.forward(idx[:sample].contiguous(), dense[:sample].contiguous(), training=True)
        err = float((engine.logit[:sample].cpu() - logit_cpu).abs().max())
        rec["parity_check"] = {"max_abs_logit_err_vs_cpu_oracle": err, "examples": sample,
                               "tolerance": 1e-5}
    # (a hipGraph that holds RCCL kernels, --graph-sharded, keeps the communicator busy: the graphs go
    # before destroy_process_group())
    if sharded:
        engine._segs = None
    del engine, idx, dense, y, batches
    import gc

    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return rec


RANK_TIMEOUT_S = float(os.environ.get("RECMAN_BENCH_TIMEOUT_S", "120"))  # collectives give up after this long


def exchange_record(engine, world, backend, B, w):
    """What a row-sharded step moves per GPU (recman_amd/dist.py): ids + rows out and gradient rows back through
    all_to_all, the dense gradients through one flat all_reduce.  (W-1)/W of it crosses xGMI."""
    n = B * w["F"]
    fixed = engine.st.capacity_factor is not None
    slots = world * engine.st.capacity(n) if fixed else n
    width = (w["D"] + 4) * 4
    a2a = slots * 8 + 2 * slots * width
    dense = int(engine._flat_grads.numel()) * 4
    remote = (world - 1) / world
    return {"world_size": world, "backend": backend,
            "layout": ("fixed-capacity buckets (equal splits, no host sync)" if fixed
                       else "dynamic split sizes (a count all_to_all + one host sync per batch)"),
            "micro_batches": engine.micro_batches, "occurrence_slots_per_gpu": slots,
            "all_to_all_bytes_per_gpu_per_step": a2a, "of_which_cross_gpu": int(a2a * remote),
            "dense_all_reduce_bytes": dense,
            "what": "ids (8 B) + fused rows [D+4] out + gradient rows [D+4] back per occurrence slot; "
                    "(W-1)/W of the all_to_all volume leaves the GPU"}


def collect(a, rank, world, local, dist, rehearse, backend, live, pmc_note, fallback):
    """Everything this rank measures -> the output record (rank 0 prints it)."""
    dev = torch.device("cuda", local)
    wname = a.workload or "deepfm"
    sharded = world > 1 or a.force_sharded
    plain = a.workload is None and not a.only and not a.batch and not a.vocab and a.zipf == 0
    extras = plain and world == 1 and not a.force_sharded
    bname = {"nccl": "nccl (RCCL over xGMI)", "gloo": "gloo (one-GPU rehearsal, host-staged)"}.get(
        backend, backend) if backend else "none (single process)"

    def sharded_sub(name, batch, steps, warm):
        """One row-sharded sub-record at the current world size (per-GPU batch `batch`)."""
        ab = argparse.Namespace(**vars(a))
        ab.batch = batch
        r = measure(ab, name, dev, rank, world, dist, rehearse, True, {}, {}, steps, warm, 0.0, want_cpu=False)
        return _sub(r, steps=steps, warmup=warm)

    rec = measure(a, wname, dev, rank, world, dist, rehearse, sharded, live, fallback, a.steps, a.warmup,
                  a.cpu_budget or 20.0, zipf=a.zipf)
    out = {
        "metric": METRIC, "value": rec.pop("value"), "unit": rec.pop("unit"), "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": rec.pop("ms_per_step"),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "world_size": (dist.get_world_size() if dist is not None else 1), "backend": bname,
    }
    out.update(rec)
    sub_steps, sub_warm = max(5, min(a.steps, 20)), max(2, min(a.warmup, 5))
    if extras:
        out["zipf"] = _sub(measure(a, "deepfm", dev, rank, world, dist, rehearse, False, {}, {}, a.steps,
                                   a.warmup, 0.0, zipf=1.05, want_cpu=False, want_opt=True),
                           note="the DeepFM workload with Zipf(1.05) ids; SURVEY.md 8d asks for both index "
                                "distributions (its optimizer_step: rows with thousands of occurrences per batch)")
        out["workloads"] = {}
        for name in ("xdeepfm", "dcn"):
            out["workloads"][name] = _sub(measure(a, name, dev, rank, world, dist, rehearse, False, live,
                                                  fallback, sub_steps, sub_warm, a.cpu_budget or 12.0),
                                          steps=sub_steps, warmup=sub_warm)
    if plain and (extras or world > 1):
        # BASELINE configs[4] - the workload the >= 6x target of north_star is about - through the row-sharded
        # engine at THIS world size (N = 1: the first point of the scaling curve), weak scaling: the per-GPU batch
        # is fixed.  Two per-GPU batches: 8192 (a global batch of 65536 over 8 GPUs) and 65536
        # (configs[1-2]'s batch on every GPU).  For N > 1 also configs[2] (xDeepFM on the 26 M-row table).
        wl = out.setdefault("workloads", {})
        s5, w5 = max(3, min(sub_steps, 10)), 2
        wl["xdeepfm_100m"] = _sub(sharded_sub("xdeepfm_100m", 8192, s5, w5),
                                  note="per-GPU batch 8192 (weak scaling; SURVEY.md 8d/8e: global 65536 at 8 GPUs)")
        wl["xdeepfm_100m_b65536"] = _sub(sharded_sub("xdeepfm_100m", 65536, s5, w5),
                                         note="per-GPU batch 65536 (weak scaling)")
        if world > 1:
            wl["xdeepfm"] = sharded_sub("xdeepfm", 0, sub_steps, sub_warm)
    if rank == 0 and world == 1:
        out["pmc_passes"] = pmc_note
    return out


def rank_main(a, rank, world, local):
    """One rank of the job (or the only process).  Fails FAST and LOUD: every collective gives up after
    RANK_TIMEOUT_S, any exception ends this rank with a non-zero code, and rank 0 prints a JSON line with an
    "error" field instead of a result - a rank that dies never leaves the others (and the driver) waiting."""
    import datetime
    import traceback

    wname = a.workload or "deepfm"
    sharded = world > 1 or a.force_sharded
    extras = (a.workload is None and not a.only and world == 1 and not a.force_sharded
              and not a.batch and not a.vocab and a.zipf == 0)
    dist = backend = None
    code = 0
    try:
        # ---- HBM traffic of the roofline kernels: rocprofv3 --pmc child passes, BEFORE any GPU call here ----
        live, pmc_note = {}, "not requested"
        full_size = not a.batch and not a.vocab and a.zipf == 0
        if rank == 0 and world == 1 and not sharded and not a.no_pmc and full_size:
            t0 = time.perf_counter()
            live, pmc_note = pmc_passes(["deepfm", "xdeepfm", "dcn"] if extras else [wname])
            pmc_note = pmc_note or f"ok ({time.perf_counter() - t0:.0f} s)"
            if not live:
                print(f"[bench] live PMC traffic unavailable: {pmc_note}", file=sys.stderr)
        fallback = json.load(open(TRAFFIC_FALLBACK)) if os.path.exists(TRAFFIC_FALLBACK) else {}

        # RECMAN_REHEARSE_ONE_GPU=1: every rank on GPU 0 with gloo (host-staged) collectives - a
        # rehearsal of the multi-rank control flow on a one-GPU box, not a measurement
        rehearse = os.environ.get("RECMAN_REHEARSE_ONE_GPU", "0") == "1"
        if rehearse:
            local = 0
        if a.selftest_ranks:
            import torch.distributed as dist

            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=RANK_TIMEOUT_S))
            kind, who, when = (a.selftest_ranks.split(":") + ["-1", "-1"])[:3]
            t = torch.zeros(1)
            for it in range(a.steps):
                if rank == int(who) and it == int(when):
                    if kind == "die":
                        os._exit(9)
                    raise RuntimeError("selftest: this rank fails here")
                t += 1
                dist.all_reduce(t)
                time.sleep(0.05)
            if rank == 0:
                print(json.dumps({"metric": METRIC, "value": float(t), "n_gpus": world, "selftest": True}), flush=True)
            dist.destroy_process_group()
            return 0
        torch.cuda.set_device(local)
        if world > 1 or "RANK" in os.environ:
            import torch.distributed as dist

            to = datetime.timedelta(seconds=RANK_TIMEOUT_S)
            if rehearse:
                dist.init_process_group("gloo", timeout=to)
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=to)
            backend = dist.get_backend()
            assert dist.get_world_size() == world
        out = collect(a, rank, world, local, dist, rehearse, backend, live, pmc_note, fallback)
        if rank == 0:
            print(json.dumps(out), flush=True)
    except BaseException as e:  # noqa: BLE001 - SystemExit included: it must still become a line and a code
        code = e.code if isinstance(e, SystemExit) and isinstance(e.code, int) and e.code else 1
        msg = f"{type(e).__name__}: {str(e)[:400]}"
        print(f"[bench] rank {rank} failed: {msg}", file=sys.stderr, flush=True)
        traceback.print_exc(file=sys.stderr)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "examples/s", "n_gpus": world,
                              "steps": a.steps, "warmup": a.warmup, "error": msg,
                              "higher_is_better": True}), flush=True)
        # no collective from here on: the peers may be gone.  os._exit skips the process-group destructor, which
        # would wait for them
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)
    if dist is not None:
        import gc

        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    return code


def main():
    a = parse()
    if a.pmc_child:
        return pmc_child(a)
    world_env = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and world_env is None:
        # not under torch.distributed.run: start the ranks ourselves, before any GPU call
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = int(world_env or 1)
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start one rank per GPU")
    sys.exit(rank_main(a, rank, world, local))


def _sub(rec, **extra):
    rec = dict(rec)
    rec.update(extra)
    return rec


if __name__ == "__main__":
    main()
