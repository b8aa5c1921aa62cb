"""profiles/<tag>_summary.md, <tag>_bench_default.json and traffic.json from gpurun_out/final_<run>/ (the output of
tools/final_profiles.sh): python3 tools/make_summary.py r02c r02"""
import json
import sys

run, tag = sys.argv[1], sys.argv[2]
out = f"gpurun_out/final_{run}/"
d = json.load(open(out + "bench_default.json"))


def table(w):
    lines = open(out + f"summary_{w}.md").read().strip().split("\n")
    return "\n".join(l for l in lines if l.startswith("|") and "torch " not in l and "__amd_rocclr" not in l)


def roofs(x):
    rs = x.get("rooflines", [x["roofline"]])
    return "; ".join(f"{r['symbol']} {r['frac']:.3f} ({r['avg_launch_us']:.1f} us"
                     + (f", {r['frac_traffic']:.3f} on PMC bytes" if r.get("frac_traffic") else "") + ")" for r in rs)


def traf(x):
    rs = x.get("rooflines", [x["roofline"]])
    return "; ".join(f"{r['traffic'] / 1e6:.0f} MB" if r.get("traffic") else "-" for r in rs)


o = d["optimizer_step"]
s = f"""# {tag} - rocprofv3 kernel summaries of the three single-GPU bench workloads (end of round 2)

`bash tools/final_profiles.sh {run}` on one MI355X: the default `python3 bench.py --steps 30 --warmup 5` line is `profiles/{tag}_bench_default.json`; per workload `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload W --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-pmc --no-optimizer` (eager launches so that every kernel is attributed; call counts include the 0.25 s of untimed pre-warm steps; torch's one-off initialisation kernels filtered out; a profiled run holds a 3-5 % lower clock than the un-profiled bench line).

## deepfm

{table('deepfm')}

## xdeepfm

{table('xdeepfm')}

## dcn

{table('dcn')}

## the bench line (profiles/{tag}_bench_default.json)

| workload | examples/s | ms/step | roofline kernels (fraction of 8 TB/s HBM or 157.3 TFLOP/s f32 MFMA; avg launch) | PMC traffic per launch | CPU port ex/s ({d['cpu_baseline']['cores']} threads) | optimizer step (reported separately) |
|---|---:|---:|---|---|---:|---:|
| deepfm (configs[1]) | {d['value']:,.0f} | {d['ms_per_step']:.4f} | {roofs(d)} | {traf(d)} | {d['cpu_baseline']['value']:,.0f} | {o['ms']:.4f} ms (sort {o['sort_ms']:.4f} + apply {o['apply_ms']:.4f}; bit-identical rerun: {o['bit_identical_rerun']}) |
| deepfm, Zipf(1.05) ids | {d['zipf']['value']:,.0f} | {d['zipf']['ms_per_step']:.4f} | {roofs(d['zipf'])} | - | - | - |
"""
for w in ("xdeepfm", "dcn"):
    x = d["workloads"][w]
    cb = x.get("cpu_baseline", {}).get("value")
    oo = x.get("optimizer_step", {}).get("ms")
    s += (f"| {w} (configs[{2 if w == 'xdeepfm' else 3}]) | {x['value']:,.0f} | {x['ms_per_step']:.4f} | {roofs(x)} | "
          f"{traf(x)} | {cb:,.0f} | {oo:.4f} ms |\n")
sh = d["step_hbm"]
s += f"""
Whole DeepFM step on SURVEY 8d's embed+FM bytes: {sh['algorithmic_bytes'] / 1e6:.1f} MB / {d['ms_per_step']:.4f} ms = {sh['achieved']:.0f} GB/s = {sh['frac']:.3f} of spec (0.362 at the start of the round; counters then: 844 MB really moved per step, profiles/r02_deepfm_step_bytes.md; the one-kernel front has since removed mlp_fwd's 123 MB re-read of E).

Round 1 -> round 2 on the same command: DeepFM 330 M -> {d['value'] / 1e6:.0f} M examples/s (one-kernel front rm_embed_mlp_fwd and the backward kernel's per-wave balance, profiles/r02_front_fusion.md; the round-2 line stores the row gradients CACHED, as fit() needs them for the optimizer step that follows - round 1's non-temporal stores, `--d-rows-reuse stream`, are worth another 2 %); xDeepFM 4.24 M -> {d['workloads']['xdeepfm']['value'] / 1e6:.2f} M (cin_fwd two chunks per barrier, cin_dw group balance); DCN 38.2 M -> {d['workloads']['dcn']['value'] / 1e6:.1f} M (dense_nn 452 -> 386 us per launch and wgrad 452 -> 410 us, profiles/r02_dense_gemm.md; cross_fwd 101 -> 48 us, cross_bwd 220 -> 93 us, profiles/r02_cross_counters.md); optimizer step 0.79 -> {o['ms']:.2f} ms (DeepFM), deterministic (profiles/r02_optimizer.md).
"""
open(f"profiles/{tag}_summary.md", "w").write(s)
json.dump(d, open(f"profiles/{tag}_bench_default.json", "w"), indent=1)
src = (f"profiles/{tag}_bench_default.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE child passes of the round-2 "
       "default bench run)")
t = {"deepfm": {}, "xdeepfm": {}, "dcn": {}}
for r in d.get("rooflines", [d["roofline"]]):
    if r.get("traffic"):
        t["deepfm"][r["symbol"]] = {"bytes": r["traffic"], "source": src}
for w in ("xdeepfm", "dcn"):
    x = d["workloads"][w]
    for r in x.get("rooflines", [x["roofline"]]):
        if r.get("traffic"):
            t[w][r["symbol"]] = {"bytes": r["traffic"], "source": src}
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
print("written")
