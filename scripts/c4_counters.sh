#!/bin/bash
# C4 step kernel: where a wave's cycles go (LDS, vector memory, waiting) -- counter passes of scripts/configs.py with CONFIGS_ONLY=C4.
# -> gpurun_out/c4ctr/*.summary.txt  (each --pmc group in a run of its own, kernel-trace off)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/c4ctr
rm -rf $OUT; mkdir -p $OUT
export CONFIGS_ONLY=C4
run_pmc() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python scripts/configs.py > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name > $OUT/$name.summary.txt 2>&1
  grep "pf_grid_step" $OUT/$name.summary.txt
}
run_pmc lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run_pmc wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
run_pmc act SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run_pmc vm SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM
run_pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
