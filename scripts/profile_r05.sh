#!/bin/bash
# Round-5 profile collection on the MI355X box (through gpurun from the repo root): kernel-trace stats of bench.py, separate
# PMC passes (HBM traffic in both weights layouts, SQ instruction mix, L2 hit / miss and memory-side request counters),
# then the other dominant kernels (C4 step kernel, C5 workgroup kernel).  Summaries land in gpurun_out/prof5/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof5
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin --no-live-pmc --no-side-runs --no-accuracy"
export BSSM_BENCH_CALIB=1
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin --no-live-pmc --no-side-runs --no-accuracy"
run_pmc() {
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- $B > $OUT/bench_stats.log 2>&1
run_pmc pmc_fetch FETCH_SIZE -- $B1
run_pmc pmc_write WRITE_SIZE -- $B1
run_pmc pmc_fetch_reflayout FETCH_SIZE -- $B1 --reference-weights-layout
run_pmc pmc_write_reflayout WRITE_SIZE -- $B1 --reference-weights-layout
run_pmc pmc_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- $B1
run_pmc pmc_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- $B1
# C4 (64 filters in flight, as bench.py runs it) and C5
C4="python scripts/c4_ab.py child 64"
run_pmc C4_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- $C4
run_pmc C4_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- $C4
run_pmc C4_sq3 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $C4
run_pmc C4_fetch FETCH_SIZE -- $C4
run_pmc C4_write WRITE_SIZE -- $C4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/C4_stats -- $C4 > $OUT/C4_stats.log 2>&1
C5="python scripts/c5_quick.py 2340 3"
run_pmc C5_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- $C5
run_pmc C5_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM -- $C5
run_pmc C5_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- $C5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/C5_stats -- $C5 > $OUT/C5_stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
find $OUT -name "*_agent_info.csv" -delete
ls $OUT
# round 5: the step kernel of sixteen-dimensional states and the matrix-core filters of 17 ... 64 dimensions
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/unified_stats -- python scripts/step_kernel_unified_probe.py > $OUT/unified_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/big_stats -- python scripts/big_states_probe.py quick > $OUT/big_stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
ls $OUT
