#!/bin/bash
# Counters of the workgroup-resident kernels at the end of round 3 (through gpurun from the repo root): C5 (psi-APF functor kernel,
# exact-size variant) and regime B of bench.py (512 filters x N = 2048, m = 4).  Summaries in gpurun_out/prof3wg/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof3wg
rm -rf $OUT; mkdir -p $OUT
run_pmc() {
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
}
C5="python scripts/c5_quick.py 2340 3"
RB="python scripts/regb_ab.py"
run_pmc C5_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- $C5
run_pmc C5_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- $C5
run_pmc C5_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- $C5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/C5_stats -- $C5 > $OUT/C5_stats.log 2>&1
run_pmc regb_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- $RB
run_pmc regb_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- $RB
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/regb_stats -- $RB > $OUT/regb_stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
grep -h "pf_wg_kernel" $OUT/*.summary.txt
