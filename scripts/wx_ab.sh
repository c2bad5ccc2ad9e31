#!/bin/bash
# weights exchange on / off in the same library (BSSM_AMD_GRID_V2_WX), interleaved: scripts/wx_ab.sh "<bench args>"
for round in 1 2; do for wx in 0 1; do echo -n "wx=$wx "; BSSM_AMD_GRID_V2_WX=$wx python bench.py --no-live-pmc --no-cpu-baseline --no-post-correct --no-regime-b --no-single --no-dropin $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g p-steps/s  %.2f ms  launch %.1f us %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernel']))"; done; done
