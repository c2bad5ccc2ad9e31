#!/bin/bash
# bench throughput under environment switches of the launcher, interleaved: scripts/env_ab.sh
run() { echo -n "$1: "; env $1 python bench.py --no-live-pmc --no-cpu-baseline --no-post-correct --no-regime-b --no-single --no-dropin --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g p-steps/s  %.2f ms %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel']))"; }
for round in 1 2; do
  run "X=0"
  run "BSSM_AMD_XCD_REMAP=0"
  run "BSSM_AMD_GRID_STREAMS=1"
  run "BSSM_AMD_GRID_V2_WX=0"
done
