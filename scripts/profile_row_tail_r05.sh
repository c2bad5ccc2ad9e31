#!/bin/bash
# kernel durations of the row family on a 37-draw tail of the C5 shape, with and without the speculative prefetch
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_row
rm -rf $OUT; mkdir -p $OUT
for s in 0 1; do
  BSSM_AMD_ROW_SPEC=$s BSSM_AMD_FORCE_PATH=row rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$s -- python scripts/c5_quick.py 37 3 > $OUT/kt$s.log 2>&1
  f=$(find $OUT/kt$s -name "*kernel_stats.csv" | head -1)
  echo "spec $s"; grep "pf_row_step\|pf_wg_kernel" "$f" | cut -c1-200
  t=$(find $OUT/kt$s -name "*kernel_trace.csv" | head -1)
  python scripts/trace_gaps.py "$t" pf_row_step
done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
