#!/usr/bin/env python
"""Headline benchmark: examples/sec of one forward+backward pass (loss -> gradient
of every parameter, optimizer step excluded) on synthetic Criteo-shaped batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload deepfm|xdeepfm|dcn]

N = 1 runs in this process; N > 1 is launched by the driver as
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (one rank
per GPU, RCCL), the embedding table row-sharded over the ranks (recman_amd/dist.py),
per-GPU batch fixed (weak scaling).  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json `configs`), seed 2019, dropout keep 1,
embedding_l2_reg 0, weights N(0, 0.01), indices uniform over each field's vocabulary:
  deepfm  (default) configs[1]: 26 sparse x 1,000,001 rows + 13 dense, D=16, MLP (32,32) relu, B=65536
  xdeepfm           configs[2]: same inputs, CIN [128,128] leaky_relu, MLP (32,32) leaky_relu
  dcn               configs[3]: 6 vector cross layers + MLP [400,400] relu, B=131072
  xdeepfm_100m      configs[4]: xDeepFM on a 100 M-row x 64-dim table (row-sharded with --gpus 8)
  dcn_matrix        (extra)    : the same with matrix cross layers x0 o (W x_l + b) + x_l
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--table-row-reuse", default="", choices=["", "stream", "cache"],
                    help="gather kernel's row loads: stream = non-temporal (default for uniform ids), "
                         "cache = plain (default with --zipf)")
    ap.add_argument("--prewarm", type=float, default=0.25,
                    help="seconds of untimed steps BEFORE the W warmup steps (GPU clock ramp); 0 = none")
    ap.add_argument("--workload", default="deepfm", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (tests)")
    ap.add_argument("--vocab", type=int, default=0, help="override rows per field (tests)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--zipf", type=float, default=0.0, help="Zipf exponent for indices (0 = uniform)")
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
    return ap.parse_args()


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


def algorithmic_bytes_embed_fwd(B, F, D, Dn, fm, lin):
    """SURVEY.md section 8d, per launch of rm_embed_fwd: idx + gathered rows (+ bias
    and linear entries) read, E (+ S, logits) written."""
    per = F * 8 + F * 4 * D + F * 4 * D  # idx, rows read, E written
    if fm:
        per += F * 4 + 4 * D + 4  # bias entries, S, fm_logit
    if lin:
        per += F * 4 + Dn * 4 + 4
    return B * per


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
                kind="port",
                sample=f"first {sample} examples of the same batch, {iters} timed fwd+bwd passes "
                       f"(median {med:.2f} s), oracle/th_layers.py on torch CPU, sparse embedding grads"
                       + (f", CIN in chunks of {chunk_max}" if chunk_max < sample else "")), logit, sample


def main():
    a = parse()
    w = dict(WORKLOADS[a.workload])
    B = a.batch or w["B"]
    V = a.vocab or w["V"]
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # RECMAN_REHEARSE_ONE_GPU=1: every rank on GPU 0 with gloo (host-staged) collectives - a
    # rehearsal of the multi-rank control flow on a one-GPU box, not a measurement
    rehearse = os.environ.get("RECMAN_REHEARSE_ONE_GPU", "0") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from recman_amd import engine as eng

    spec = eng.FeatureSpec([f"C{i + 1}" for i in range(w["F"])], [V] * w["F"],
                           [f"I{j + 1}" for j in range(w["Dn"])])
    hp = dict(w["hp"], embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_l2_reg=0.0, cin_l2_reg=0.0,
              cross_layer_l2_reg=0.0,
              # uniform ids: a row is touched about once per batch -> streamed (non-temporal) row loads;
              # --zipf: hot rows want the caches
              table_row_reuse=a.table_row_reuse or ("cache" if a.zipf > 0 else "stream"))
    sharded = world > 1 or a.force_sharded
    if sharded:
        from recman_amd import dist as rdist

        fixed = a.exchange == "fixed" or (a.exchange == "auto" and a.zipf == 0)
        micro = a.micro_batches or (2 if world > 1 else 1)
        engine = rdist.make_sharded_engine(w["model"], spec, w["D"], hp, dev, rank, world,
                                           capacity_factor=1.0 if fixed else None, micro_batches=micro)
    else:
        engine = eng.ENGINES[w["model"]](spec, w["D"], hp, device=dev)
    init_engine(engine, 2019)
    idx, dense, y = synth_inputs(w, B, V, dev, 2019 + rank, a.zipf)

    def step():
        return engine.fwd_bwd(idx, dense, y)

    # ---- optional hipGraph capture of the whole step (launch-bound otherwise) ----
    # (the dynamic exchange needs host-side split sizes and cannot be captured)
    fixed = sharded and engine.st.capacity_factor is not None
    use_graph = not a.no_graph and (not sharded or (fixed and a.graph_sharded))
    graph = None
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

            graph = torch.cuda.CUDAGraph()
            gc.collect()
            gc.disable()  # (a collection inside the capture could free device memory: unsafe there)
            try:
                with torch.cuda.graph(graph):
                    step()
            finally:
                gc.enable()
        except Exception as e:  # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture failed, running eager: {e}", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    segments = False
    if sharded and fixed and graph is None and not a.no_graph and not a.no_graph_segments:
        # the compute between the collectives as hipGraphs, the RCCL calls eager in between
        try:
            engine.capture_segments(idx, dense, y)
            segments = True
        except Exception as e:
            print(f"[bench] segment capture failed, running eager: {e}", file=sys.stderr)
            engine._segs = None
            torch.cuda.synchronize()
    run = graph.replay if graph is not None else step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: a step takes 0.2 ms, so W warmup steps alone end before the GPU has left its idle
    # clocks (50 timed steps measured 2.5 % slower than 1000).  Untimed, before the W warmup steps.
    if a.prewarm > 0 and sharded:
        # (a FIXED count with collectives inside: every rank must issue the same number of them)
        for _ in range(100):
            run()
        torch.cuda.synchronize()
    elif a.prewarm > 0:
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < a.prewarm:
            for _ in range(20):
                run()
            torch.cuda.synchronize()
    for _ in range(a.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    enqueue = time.perf_counter() - t0  # host time to enqueue the steps (launch-bound when ~= elapsed)
    barrier()
    elapsed = time.perf_counter() - t0
    if any_overflow():
        raise SystemExit("[bench] a timed batch overflowed the fixed-capacity exchange: rerun with --exchange dynamic")
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms = elapsed / a.steps * 1e3
    value = world * B * a.steps / elapsed

    # ---- roofline of the dominant hand-written kernel, HIP events on its stream ----
    # (sharded: only for the models whose dominant kernel is the gather; rank 0, local kernel only)
    roof = None
    if not sharded:
        roof = engine.roofline_probe(idx, dense, y, iters=max(10, min(a.steps, 50)))
    elif rank == 0 and w["model"] == "deepfm":
        roof = engine.roofline_probe(idx, dense, y, iters=max(10, min(a.steps, 50)))
    if roof is not None and not sharded and a.workload == "deepfm" and B == 65536 and V == 1_000_001 and a.zipf == 0:
        # HBM bytes per launch of the gather kernel from rocprofv3 PMC passes of this same command
        # (FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes on gfx950, + WRITE_SIZE):
        # profiles/r01_p5_deepfm_fused.md
        roof["traffic"] = 375.0e6
        roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, profiles/r01_p5_deepfm_fused.md"
    if (roof is not None and not sharded and B == w["B"] and V == 1_000_001 and a.zipf == 0
            and a.workload in ("xdeepfm", "dcn")):
        # HBM bytes per launch of the dominant MFMA kernel (same PMC recipe): profiles/r01_p10_traffic.md
        roof["traffic"] = {"xdeepfm": 475.0e6 + 591.0e6, "dcn": 249.0e6 + 217.0e6}[a.workload]
        roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE (x2) / --pmc WRITE_SIZE, profiles/r01_p10_traffic.md"

    out = {
        "metric": "examples/sec fwd+bwd, Criteo-shape batch 65536; % HBM and MFMA roofline",
        "value": round(value, 1), "unit": "examples/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"{a.workload} (BASELINE configs[{1 + ['deepfm', 'xdeepfm', 'dcn'].index(a.workload)}])"
                                if a.workload in ("deepfm", "xdeepfm", "dcn") else
                                "xdeepfm_100m (BASELINE configs[4])" if a.workload == "xdeepfm_100m" else
                                f"{a.workload} (extra workload, not a BASELINE config)"),
                   "batch_per_gpu": B, "batch_all_gpus": B * world, "sparse_fields": w["F"],
                   "rows_per_field": V,
                   "dense_fields": w["Dn"], "emb_dim": w["D"], "hp": {k: v for k, v in w["hp"].items()},
                   "indices": "uniform" if a.zipf == 0 else f"zipf({a.zipf})",
                   "table_row_loads": ("cached" if hp["table_row_reuse"] == "cache"
                                       else "non-temporal (ids with little reuse per batch)"),
                   "embedding_l2_reg": 0.0, "dropout_keep": 1.0, "optimizer_step": "excluded",
                   "host_enqueue_ms_per_step": round(enqueue / a.steps * 1e3, 4),
                   "prewarm_s": a.prewarm,
                   "hipgraph": ("segments between the collectives" if segments else graph is not None),
                   "table": (f"row-sharded mod {world}, fused [D+4] rows, all_to_all over xGMI, "
                             + ("fixed-capacity exchange (equal splits, no host sync)" if fixed
                                else "dynamic split sizes (one host sync per batch)")
                             + f", {engine.micro_batches} micro-batch(es) per step") if sharded
                   else "single GPU"},
        "roofline": roof,
    }
    if rank == 0 and not sharded:
        # reported separately (the headline metric is fwd+bwd only): the lazy row-wise Adam step
        # on the table rows this batch touched + dense Adam on the dense parameters
        try:
            from recman_amd.optim import Optimizer, SparseTableOptimizer

            sopt, dopt = SparseTableOptimizer(engine, "adam", 1e-3), Optimizer("adam", 1e-3)
            for _ in range(3):
                sopt.step(idx)
                dopt.step(engine.params, engine.grads)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(10):
                sopt.step(idx)
                dopt.step(engine.params, engine.grads)
            ev1.record()
            torch.cuda.synchronize()
            out["optimizer_step"] = {"ms": round(ev0.elapsed_time(ev1) / 10, 4),
                                     "what": "lazy row-wise Adam on touched table rows (rm_sparse_optimizer_step) "
                                             "+ dense Adam on dense parameters; NOT part of value"}
            del sopt, dopt
        except Exception as e:  # never let the extra break the contract line
            out["optimizer_step"] = {"ms": None, "error": str(e)[:200]}
    if rank == 0 and not sharded and not a.no_cpu_baseline:
        base, logit_cpu, sample = cpu_baseline(w, hp, idx, dense, y, engine)
        out["cpu_baseline"] = base
        engine.forward(idx[:sample].contiguous(), dense[:sample].contiguous(), training=True)
        err = float((engine.logit[:sample].cpu() - logit_cpu).abs().max())
        out["parity_check"] = {"max_abs_logit_err_vs_cpu_oracle": err, "examples": sample,
                               "tolerance": 1e-5}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        # graphs first: a hipGraph that holds RCCL kernels (--graph-sharded) keeps the communicator
        # busy and destroy_process_group() waited on it forever
        run = graph = None
        if sharded:
            engine._segs = None
        import gc

        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
