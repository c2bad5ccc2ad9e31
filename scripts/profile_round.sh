#!/bin/bash
# Round profile collection on the MI355X box (run through gpurun from the repo root):
#   kernel-trace stats of bench.py and of scripts/configs.py, then separate PMC passes of bench.py.
# Summaries land in gpurun_out/prof/; copy what is judged into profiles/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-live-pmc --steps 5 --warmup 1 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin"
export BSSM_BENCH_CALIB=1
B1="python bench.py --no-live-pmc --steps 1 --warmup 0 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- $B > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs_stats -- python scripts/configs.py > $OUT/configs_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- $B1 > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq2 -- $B1 > $OUT/pmc_sq2.log 2>&1
for d in pmc_fetch pmc_write pmc_sq1 pmc_sq2; do python scripts/pmc_summary.py $OUT/$d > $OUT/$d.summary.txt 2>&1; done
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
# keep only the summaries (raw traces are large)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
ls -la $OUT
