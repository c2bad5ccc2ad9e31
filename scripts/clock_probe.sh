#!/bin/bash
# Engine clock and power while the kernels run (through gpurun from the repo root): rocm-smi sampled every second during the bench batch, C4 and the 8x C5 job
mkdir -p gpurun_out/clk
L=gpurun_out/clk/samples.txt; : > $L
( while true; do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use" | sed -E 's/.*: //' | tr '\n' ' ')" >> $L; sleep 0.7; done ) & S=$!
mark() { echo "$(date +%s.%N | cut -c1-14) == $1" >> $L; }
mark "bench batch start"; python bench.py --no-live-pmc --steps 300 --warmup 2 --no-cpu-baseline --no-single --no-regime-b --no-post-correct --no-dropin --no-accuracy --no-mng > gpurun_out/clk/b.json 2>/dev/null; mark "bench batch end"
mark "C4 start"; for i in 1 2 3; do python scripts/c4_ab.py child 64; done > gpurun_out/clk/c4.txt 2>/dev/null; mark "C4 end"
mark "C5x8 start"; python scripts/c5_quick.py 18720 40 > gpurun_out/clk/c5.txt 2>/dev/null; mark "C5x8 end"
kill $S
grep -E "==|\(2[0-9]{3}Mhz\)|\(1[0-9]{3}Mhz\)" $L
