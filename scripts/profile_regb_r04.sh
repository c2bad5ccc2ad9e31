#!/bin/bash
# Regime B of bench.py (512 workgroup-resident filters x N = 2048, m = 4) after the late-round-4 occupancy change: kernel stats and
# SQ counters -> gpurun_out/prof_regb/ (copied to profiles/r04_pmc_regb_v2.txt, r04_regb_kernel_stats.csv)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_regb
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 1 --warmup 0 --replicas 2 --no-cpu-baseline --no-single --no-post-correct --no-dropin --no-live-pmc --no-side-runs --no-accuracy --no-mng"
run_pmc() {
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
  grep pf_wg_kernel $OUT/$name.summary.txt
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/regb_stats -- $B > $OUT/regb_stats.log 2>&1
run_pmc regb_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- $B
run_pmc regb_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- $B
run_pmc regb_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -- $B
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/regb_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
grep "pf_wg_kernel" $OUT/regb_kernel_stats.csv | head -3
