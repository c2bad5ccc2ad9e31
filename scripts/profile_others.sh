#!/bin/bash
# Counters for the dominant kernels of the configurations OTHER than the headline (through gpurun from the repo root):
#   C4  pf_grid_step<8,8,1>   (ssm_mng m = 8, fp64-MFMA transition)        CONFIGS_ONLY=C4 scripts/configs.py
#   C5  pf_wg_kernel<2,2,...> (ssm_nlg growth psi-APF, state in LDS)       CONFIGS_ONLY=C5
#   regime B pf_wg_kernel<4,4,4,false,512>                                 bench.py regime_b leg
# PMC passes are separate runs with nothing but --pmc (kernel-trace stats in their own pass).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_others
rm -rf $OUT; mkdir -p $OUT
run_pmc() {  # name, counters..., then -- command
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
}
for cfg in C4 C5; do
  export CONFIGS_ONLY=$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${cfg}_stats -- python scripts/configs.py > $OUT/${cfg}_stats.log 2>&1
  run_pmc ${cfg}_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- python scripts/configs.py
  run_pmc ${cfg}_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- python scripts/configs.py
  run_pmc ${cfg}_sq3 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python scripts/configs.py
  run_pmc ${cfg}_fetch FETCH_SIZE -- python scripts/configs.py
  run_pmc ${cfg}_write WRITE_SIZE -- python scripts/configs.py
done
unset CONFIGS_ONLY
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single --no-post-correct"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/regb_stats -- $B > $OUT/regb_stats.log 2>&1
run_pmc regb_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- $B
run_pmc regb_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- $B
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/*.summary.txt | grep -i "pf_wg\|pf_grid_step"
