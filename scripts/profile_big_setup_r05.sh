#!/bin/bash
# Kernel statistics and LDS / instruction counters of the psi set-up of bsm_ng(period 24 / 52) draws (states padded to 32 / 64:
# ung_smooth_wave_kernel, ung_factors_rows_kernel) and the bench line's kernel statistics at the end of round 5.  Summaries in gpurun_out/prof5s/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof5s
rm -rf $OUT; mkdir -p $OUT
P="python scripts/big_setup_probe.py 256"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/big_setup_stats -- $P > $OUT/big_setup_stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU --output-format csv -d $OUT/big_setup_sq1 -- $P > $OUT/big_setup_sq1.log 2>&1
python scripts/pmc_summary.py $OUT/big_setup_sq1 > $OUT/big_setup_sq1.summary.txt 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/big_setup_lds -- $P > $OUT/big_setup_lds.log 2>&1
python scripts/pmc_summary.py $OUT/big_setup_lds > $OUT/big_setup_lds.summary.txt 2>&1
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin --no-live-pmc --no-side-runs --no-accuracy"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- $B > $OUT/bench_stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
find $OUT -name "*_agent_info.csv" -delete
ls -la $OUT
