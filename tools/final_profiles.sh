#!/bin/bash
# Regenerates a round's bench line and rocprofv3 kernel summaries on the GPU box:
#   gpurun -- 'bash tools/final_profiles.sh r03'   then copy gpurun_out/final_r03/* into profiles/.
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final_$tag
mkdir -p $out
python3 bench.py --steps 30 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
for w in deepfm xdeepfm dcn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -- python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-pmc --no-optimizer > /dev/null 2>&1
  python3 tools/prof_summary.py $out/prof_$w 30 > $out/summary_$w.md
done
bash tools/pmc_kernel.sh ${tag}_cross tools/bench_cross.py > /dev/null 2>&1 || true
bash tools/pmc_kernel.sh ${tag}_optim tools/bench_optim.py > /dev/null 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_optim_zipf -- python3 tools/bench_optim.py 1.05 > /dev/null 2>&1 || true
python3 tools/prof_summary.py $out/prof_optim_zipf 12 > $out/summary_optim_zipf.md 2>/dev/null || true
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_sharded -- python3 bench.py --workload deepfm --only --force-sharded --no-graph --no-graph-segments --no-optimizer --no-cpu-baseline --no-pmc --steps 20 > /dev/null 2>&1 || true
python3 tools/prof_summary.py $out/prof_sharded 15 > $out/summary_sharded_deepfm.md 2>/dev/null || true
cp gpurun_out/${tag}_cross/summary.md $out/pmc_cross.md 2>/dev/null || true
cp gpurun_out/${tag}_optim/summary.md $out/pmc_optim.md 2>/dev/null || true
echo done
