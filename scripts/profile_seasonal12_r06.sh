#!/bin/bash
# Round 6: kernel statistics of a post-correction of 2 048 bsm_ng(period = 12) draws (13 states, padded to 16: lg_smoother_rows.h, LgFactorRows<16>).
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof6m
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s12 -- python $ROOT/scripts/seasonal_probe.py 2048 ${PERIOD:-12} > $OUT/s12.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/s12_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
head -12 $OUT/s12_kernel_stats.csv | cut -c1-150; tail -3 $OUT/s12.log
