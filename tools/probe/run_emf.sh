cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_front.py -x -q 2>&1 | tail -2
for v in p e0 e1; do
  if [ $v = p ]; then unset RECMAN_HIP_LIB; else export RECMAN_HIP_LIB=build/librecman_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emf$v -- python3 bench.py --workload deepfm --only --steps 30 --warmup 5 --no-cpu-baseline --no-graph --no-pmc --no-optimizer > /dev/null 2>&1
  echo "== $v"; python3 tools/prof_summary.py gpurun_out/prof_emf$v 2 | grep "embed_mlp\|mlp_bwd"
done
