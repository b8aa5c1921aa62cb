"""Per-kernel averages of the counters in a rocprofv3 --pmc run directory.
usage: python tools/pmc_summary.py <dir> [kernel-name-substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0][-48:]
        if flt and flt not in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:36s} calls {len(v):4d}  avg {sum(v) / len(v):.4g}")
    g = cs.get("GRBM_GUI_ACTIVE")
    m = cs.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if g and m:
        ga, ma = sum(g) / len(g), sum(m) / len(m)
        print(f"   -> MfmaUtil = {ma / (ga / 8 * 1024):.3f}")
    w, wc = cs.get("SQ_WAIT_ANY"), cs.get("SQ_WAVE_CYCLES")
    if w and wc:
        print(f"   -> wait = {sum(w) / sum(wc):.3f}")
