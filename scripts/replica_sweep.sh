#!/bin/bash
# particle-steps/s of the bench workload by filters in flight (and particles per lane): scripts/replica_sweep.sh
for ppt in 2 4; do for r in 16 24 32 40 48 64; do
  echo -n "ppt $ppt replicas $r: "
  BSSM_AMD_GRID_PPT=$ppt python bench.py --no-live-pmc --no-cpu-baseline --no-post-correct --no-regime-b --no-single --no-dropin --steps 3 --warmup 1 --replicas $r 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g p-steps/s  %.2f ms  launch %.1f us  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernel']))"
done; done
