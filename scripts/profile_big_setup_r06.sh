#!/bin/bash
# Round 6: kernel statistics of the psi set-up of bsm_ng(period 24 / 52) draws after the thin conditional factors (lg_factors_rows.h).
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof6s
rm -rf $OUT; mkdir -p $OUT
P="python $ROOT/scripts/big_setup_probe.py 256"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/big_setup_stats -- $P > $OUT/big_setup_stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do cp $f $OUT/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
ls -la $OUT
