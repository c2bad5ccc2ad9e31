#!/bin/bash
# rocprofv3 kernel trace of one rank's 293-draw shard of the C5 post-correction (through gpurun from the repo root):
# per-kernel durations of the full round (pf_wg_kernel) and of the tail (pf_row_step) -> profiles/r05_c5_shard_kernel_stats.csv
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_c5s
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python scripts/c5_quick.py 293 4 > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-260
tail -3 $OUT/kt.log
find $OUT -name "*.db" -delete; find $OUT/kt -name "*kernel_trace.csv" -size +20M -delete
