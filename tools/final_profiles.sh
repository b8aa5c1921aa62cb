#!/bin/bash
# Regenerates the round's final bench lines and rocprofv3 kernel summaries on the GPU box:
#   gpurun -- 'bash tools/final_profiles.sh'   then copy gpurun_out/final/* into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final
mkdir -p $out
for w in deepfm xdeepfm dcn; do
  python bench.py --workload $w --steps 30 --warmup 5 > $out/bench_$w.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -- python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-graph > /dev/null 2>&1
  python tools/prof_summary.py $out/prof_$w 30 > $out/summary_$w.md
done
python bench.py --zipf 1.05 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_deepfm_zipf.json
echo done
