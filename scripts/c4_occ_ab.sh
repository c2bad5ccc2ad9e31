cp bssm_amd/libbssm_amd.so /tmp/keep.so
for r in 1 2 3; do
  cp /tmp/keep.so bssm_amd/libbssm_amd.so; echo -n "base ppt2 occ4: "; python scripts/ab_probe.py c4_64,c4_32 2>/dev/null
  cp /tmp/keep.so bssm_amd/libbssm_amd.so; echo -n "base ppt1 occ4: "; BSSM_AMD_GRID_PPT=1 python scripts/ab_probe.py c4_64,c4_32 2>/dev/null
  cp ab_c4occ5.so bssm_amd/libbssm_amd.so; echo -n "occ5 ppt1:      "; BSSM_AMD_GRID_PPT=1 python scripts/ab_probe.py c4_64,c4_32 2>/dev/null
done
cp /tmp/keep.so bssm_amd/libbssm_amd.so
