#!/bin/bash
# A/B of library builds (ab_*.so at the repo root) on one GPU box, interleaved: scripts/ab2.sh [what] -> gpurun_out/ab2.txt
cp bssm_amd/libbssm_amd.so /tmp/keep.so
for round in $(seq 1 ${AB_ROUNDS:-3}); do for f in ab_*.so; do cp $f bssm_amd/libbssm_amd.so; echo -n "$f "; python scripts/ab_probe.py $1 2>/dev/null; done; done | tee gpurun_out/ab2.txt
cp /tmp/keep.so bssm_amd/libbssm_amd.so
