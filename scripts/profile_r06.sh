#!/bin/bash
# Round-6 profile collection on the MI355X box (through gpurun from the repo root): kernel-trace stats of the DEFAULT bench command (the timed legs as
# the driver runs them, `value` on the reference weights layout since this round), the HBM-traffic counter passes of the headline kernel in both
# weights layouts, the SQ instruction counters.  Summaries land in gpurun_out/prof6/.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof6
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin --no-live-pmc --no-live-valu --no-side-runs --no-accuracy --no-mng --no-clock"
export BSSM_BENCH_CALIB=1
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin --no-live-pmc --no-live-valu --no-side-runs --no-accuracy --no-mng --no-clock"
run_pmc() {
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- $B > $OUT/bench_stats.log 2>&1
run_pmc pmc_fetch FETCH_SIZE -- $B1
run_pmc pmc_write WRITE_SIZE -- $B1
run_pmc pmc_fetch_factored FETCH_SIZE -- $B1 --factored-weights-layout
run_pmc pmc_write_factored WRITE_SIZE -- $B1 --factored-weights-layout
run_pmc pmc_sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -- $B1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
find $OUT -name "*_agent_info.csv" -delete
grep -h "bench_stats" /dev/null; tail -2 $OUT/bench_stats.log; head -5 $OUT/bench_stats_kernel_stats.csv | cut -c1-200; grep -h "pf_grid_step\|calib" $OUT/pmc_*.summary.txt
