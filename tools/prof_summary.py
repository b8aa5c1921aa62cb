#!/usr/bin/env python
"""Condenses a rocprofv3 `--kernel-trace --stats --output-format csv` run into a short
markdown table (kernel names shortened) - the form committed under profiles/."""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]{0,40}>)?)", name)
    s = m.group(1) if m else name
    if s.startswith("Cijk_"):
        mt = re.search(r"MT(\d+x\d+x\d+)", name)
        s = "hipBLASLt " + name[:14] + ("_MT" + mt.group(1) if mt else "")
    if s.startswith("at::native"):
        f = re.search(r"(\w+Functor|\w+_kernel_cuda|sum_functor|launch_clamp_scalar|compare_scalar_kernel|normal_kernel|random_from_to_kernel|uniform_kernel)", name)
        s = "torch " + s.split("::")[2].split("<")[0] + ("/" + f.group(1) if f else "")
    return s[:90]


def main(d, steps=None, top=25):
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))
    if not f:
        sys.exit("no *kernel_stats.csv under " + d)
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for r in rows[:top]:
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
              f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    print(f"\ntotal kernel time {tot/1e6:.3f} ms over {len(rows)} distinct kernels")


if __name__ == "__main__":
    main(sys.argv[1], top=int(sys.argv[2]) if len(sys.argv) > 2 else 25)
