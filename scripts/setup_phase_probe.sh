#!/bin/bash
# Phase times of fn_setup_kernel (psi set-up of ssm_nlg post-correction chunks), workgroup 0: a library built with
# -DBSSM_SETUP_STAMPS (`make -C bssm_amd/csrc stamps` -> bssm_amd/ab_stamps.so, built before the gpurun call) prints them from
# the kernel.  Through gpurun from the repo root.
set -e
cp bssm_amd/libbssm_amd.so /tmp/lib_keep.so
cp bssm_amd/ab_stamps.so bssm_amd/libbssm_amd.so
for ns in 2340 292; do echo "== $ns draws"; python scripts/c5_quick.py $ns 2 2>&1 | grep -E "fn_setup_kernel|seconds" | tail -4; done
echo "== ssm_mng C4 shape, 512 draws"; python scripts/mng_one_call.py 2>&1 | grep -E "mng_setup_kernel|call" | tail -4
cp /tmp/lib_keep.so bssm_amd/libbssm_amd.so
