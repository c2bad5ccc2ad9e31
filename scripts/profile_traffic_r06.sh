#!/bin/bash
# Round 6: HBM bytes per launch of pf_big_row_step (25 / 53 states, 256 filters x 1024 particles, log-likelihood only) and of the C4 step kernel
# (64 filters x 16 384 particles, full output): separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH factor calibrated in the same process on
# calib_copy_kernel (MI355X_MICROARCH.md, HBM section).  Summaries -> gpurun_out/traf6/*.txt
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/traf6
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for c in big24 big52 c4; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/${c}_$ctr -- python $ROOT/scripts/traffic_cases_r06.py $c > $OUT/${c}_$ctr.log 2>&1
    python $ROOT/scripts/pmc_summary.py $OUT/${c}_$ctr > $OUT/${c}_$ctr.summary.txt 2>&1
  done
done
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
cat $OUT/*.summary.txt | grep -E "pf_big_row_step|pf_grid_step|calib_copy"
