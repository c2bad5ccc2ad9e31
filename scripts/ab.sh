#!/bin/bash
# A/B two builds of libbssm_amd.so on one GPU box: scripts/ab.sh "<bench args>"   (expects ab_A.so, ab_B.so at the repo root)
for v in A B A B; do cp ab_$v.so bssm_amd/libbssm_amd.so; echo -n "$v "; python bench.py --no-cpu-baseline --no-post-correct --no-regime-b --no-single $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g p-steps/s  %.2f ms' % (d['value'], d['ms_per_step']))"; done
