#!/bin/bash
# What bounds the step kernel: the same batch with and without the weights output (24 of its 111 HBM bytes per
# particle-step), at several batch sizes.  Through gpurun from the repo root.
B="python bench.py --no-live-pmc --no-cpu-baseline --no-post-correct --no-regime-b --no-single"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g p-steps/s  %.2f ms' % (d['value'], d['ms_per_step']))"; }
for R in 16 24 32 48; do echo -n "replicas $R            "; $B --replicas $R 2>/dev/null | show; done
for R in 24 32; do echo -n "replicas $R no weights "; BSSM_AMD_PROBE_NO_WEIGHTS=1 $B --replicas $R 2>/dev/null | show; done
for P in 2 4; do echo -n "replicas 24 ppt $P      "; BSSM_AMD_GRID_PPT=$P $B 2>/dev/null | show; done
echo -n "replicas 24 one stream "; BSSM_AMD_GRID_STREAMS=1 $B 2>/dev/null | show
