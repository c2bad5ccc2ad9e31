#!/bin/bash
# Quick PMC look at the bench kernels (through gpurun from the repo root): instruction counts and busy cycles.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/profq
rm -rf $OUT; mkdir -p $OUT
B1="python bench.py --no-live-pmc --steps 1 --warmup 0 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- $B1 > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq2 -- $B1 > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq3 -- $B1 > $OUT/pmc_sq3.log 2>&1
for d in pmc_sq1 pmc_sq2 pmc_sq3; do python scripts/pmc_summary.py $OUT/$d > $OUT/$d.summary.txt 2>&1; done
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/*.summary.txt | grep -i "pf_grid"
