#!/bin/bash
# Kernel stats + counters of ONE configuration of scripts/configs.py (default C4; scripts/profile_c4.sh C5 for the post-correction kernel):
# scripts/profile_others.sh on its own (through gpurun from the repo root).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_cfg
rm -rf $OUT; mkdir -p $OUT
run_pmc() {
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
}
CFG=${1:-C4}
export CONFIGS_ONLY=$CFG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${CFG}_stats -- python scripts/configs.py > $OUT/${CFG}_stats.log 2>&1
run_pmc ${CFG}_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- python scripts/configs.py
run_pmc ${CFG}_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- python scripts/configs.py
run_pmc ${CFG}_sq3 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python scripts/configs.py
run_pmc ${CFG}_fetch FETCH_SIZE -- python scripts/configs.py
run_pmc ${CFG}_write WRITE_SIZE -- python scripts/configs.py
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/*.summary.txt | grep -i "pf_grid_step\|pf_wg_kernel"
