for rnd in 1 2; do
echo "== global path"; BSSM_AMD_SS_GLOBAL=1 python scripts/c5_outputs_probe.py 2>&1 | grep -E "output 2 threads  8"
echo "== register path, nt loads"; python scripts/c5_outputs_probe.py 2>&1 | grep -E "output 2 threads  8"
cp bssm_amd/libbssm_amd.so /tmp/keep.so; cp ssnt0.so bssm_amd/libbssm_amd.so
echo "== register path, plain loads"; python scripts/c5_outputs_probe.py 2>&1 | grep -E "output 2 threads  8"
cp /tmp/keep.so bssm_amd/libbssm_amd.so
done
