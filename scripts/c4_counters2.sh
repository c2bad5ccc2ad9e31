#!/bin/bash
# C4 step kernel, second set: instruction fetch / instruction cache, barrier and scalar-memory waits, occupancy
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/c4ctr2
rm -rf $OUT; mkdir -p $OUT
export CONFIGS_ONLY=C4
run_pmc() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python scripts/configs.py > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
  grep "pf_grid_step" $OUT/$name.summary.txt || tail -3 $OUT/$name.log
}
run_pmc ifetch SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run_pmc icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run_pmc dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM
run_pmc bar SQ_INSTS_WAVE32_LDS SQ_WAVES_LT_64 SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES
run_pmc misc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|SQC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/avail_sq.txt
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
