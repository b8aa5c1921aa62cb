#!/bin/bash
# Counter evidence for one micro-benchmark script: kernel trace + stats, then PMC passes (each counter
# group in its own run, kernel-trace only - the combination gpurun allows).
#   bash tools/pmc_kernel.sh <tag> <python script> [args...]      -> gpurun_out/<tag>/{stats,sq,fetch,write,tcc}
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 "$@" > $out/run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/sq -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $out/sq2 -- python3 "$@" > /dev/null 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/tcc -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/mfma -- python3 "$@" > /dev/null 2>&1 || true
python3 tools/prof_summary.py $out/stats 12 > $out/summary.md
for d in sq sq2 fetch write tcc mfma; do echo "## $d" >> $out/summary.md; python3 tools/pmc_summary.py $out/$d >> $out/summary.md 2>&1 || true; done
cat $out/summary.md
