#!/bin/bash
# Round-3 profile collection on the MI355X box (through gpurun from the repo root): kernel-trace stats of bench.py, separate
# PMC passes (HBM traffic in both weights layouts, SQ instruction mix, L2 hit / miss and memory-side request counters),
# then the other dominant kernels (C4 step kernel, C5 workgroup kernel).  Summaries land in gpurun_out/prof3/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof3
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-live-pmc --steps 5 --warmup 1 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin"
export BSSM_BENCH_CALIB=1
B1="python bench.py --no-live-pmc --steps 1 --warmup 0 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin"
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
# what the profiler offers for the caches behind the L2 (recorded as found; a pass whose counters do not exist just fails)
rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|EA0_R|EA0_W|TCC_HIT|TCC_MISS|TCC_REQ" | head -80 > $OUT/avail_memory_counters.txt
run_pmc pmc_tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $B1
run_pmc pmc_tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -- $B1
run_pmc pmc_tcc3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum -- $B1
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
