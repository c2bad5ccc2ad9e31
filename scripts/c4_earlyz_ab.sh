#!/bin/bash
# same-box A/B of the C4 step kernel with the transition normals drawn under the search's round trips (BSSM_V2_EARLY_Z = 0 ... 4,
# pf_grid_v2.h): ab_ez<level>.so built from pf_inst_mp8.hip with -DBSSM_V2_EARLY_Z=<level> -> gpurun_out/c4_earlyz_ab.txt
cp bssm_amd/libbssm_amd.so /tmp/keep.so
{
for r in 1 2 3; do
  for v in ez0 ez1 ez2 ez3 ez4; do cp ab_$v.so bssm_amd/libbssm_amd.so; echo -n "$v: "; python scripts/ab_probe.py c4_64,c4_32 2>/dev/null; done
done
for v in ez1 ez2 ez3 ez4; do
  cp ab_$v.so bssm_amd/libbssm_amd.so; echo -n "$v parity: "
  python -m pytest tests/test_gpu_mng.py tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED"
done
} | tee gpurun_out/c4_earlyz_ab.txt
cp /tmp/keep.so bssm_amd/libbssm_amd.so
