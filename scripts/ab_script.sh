#!/bin/bash
# A/B builds of libbssm_amd.so with any script: scripts/ab_script.sh "<python script + args>"   (expects ab_*.so at the
# repo root; every variant is run twice, interleaved)
cp bssm_amd/libbssm_amd.so /tmp/keep.so
for round in 1 2; do for f in ab_*.so; do cp $f bssm_amd/libbssm_amd.so; echo "== $f"; python $1 2>&1 | tail -${2:-5}; done; done
cp /tmp/keep.so bssm_amd/libbssm_amd.so
