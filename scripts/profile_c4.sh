#!/bin/bash
# Kernel stats + counters of configuration C4 (ssm_mng m = 8 p = 4, pf_grid_step_v2<8,8,.,true>): the C4 part of
# scripts/profile_others.sh on its own (through gpurun from the repo root).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_c4
rm -rf $OUT; mkdir -p $OUT
run_pmc() {
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
}
export CONFIGS_ONLY=C4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/C4_stats -- python scripts/configs.py > $OUT/C4_stats.log 2>&1
run_pmc C4_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- python scripts/configs.py
run_pmc C4_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- python scripts/configs.py
run_pmc C4_sq3 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python scripts/configs.py
run_pmc C4_fetch FETCH_SIZE -- python scripts/configs.py
run_pmc C4_write WRITE_SIZE -- python scripts/configs.py
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/*.summary.txt | grep -i "pf_grid_step"
