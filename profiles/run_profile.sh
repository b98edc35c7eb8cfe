#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's dominant kernel on the GPU box.
#   usage (from the repo root, on the box):  bash profiles/run_profile.sh <tag> [bench args...]
# Writes raw output under gpurun_out/prof_<tag>/ (scratch); summarize.py turns it into the
# committed profiles/<tag>_*.csv/json.  Kernel-trace/stats and each PMC group are SEPARATE
# runs (gpurun refuses --pmc together with trace domains other than kernel-trace/stats).
set -o pipefail
#   RT_PROFILE_PROG=tools/bench_bloom.py (or tools/bench_taa.py) profiles a secondary chain instead of bench.py.
TAG=${1:-r01}; shift
PROG=${RT_PROFILE_PROG:-bench.py}
if [ "$PROG" = bench.py ]; then ARGS=${@:---steps 30 --warmup 5 --no-cpu-baseline --extra-configs none --no-modes}; else ARGS=$@; fi
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}
mkdir -p $OUT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $PROG $ARGS > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_FLAT"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $PROG $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc group failed: $grp"; tail -5 $OUT/pmc_$name.log; }
done
python3 profiles/summarize.py $OUT profiles/${TAG}
