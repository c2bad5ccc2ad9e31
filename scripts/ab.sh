#!/bin/bash
# A/B builds of libbssm_amd.so on one GPU box: scripts/ab.sh "<bench args>"   (expects ab_*.so at the repo root; every
# variant is run twice, interleaved)
cp bssm_amd/libbssm_amd.so /tmp/keep.so
for round in $(seq 1 ${AB_ROUNDS:-2}); do for f in ab_*.so; do cp $f bssm_amd/libbssm_amd.so; echo -n "$f "; python bench.py --no-live-pmc --no-cpu-baseline --no-post-correct --no-regime-b --no-single --no-dropin $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g p-steps/s  %.2f ms  launch %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))"; done; done
cp /tmp/keep.so bssm_amd/libbssm_amd.so
