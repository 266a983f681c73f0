#!/bin/bash
# PMC passes over the fused kernel (counters only: no trace domains besides kernel-trace).  Usage: [RSIM_CONFIG=stack|peg|pickplace] tools/pmc_pass.sh <tag> [sets...]
tag=${1:-pmc}; shift
sets=${@:-sq1 sq2 ic}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { # name, counters
  rm -rf gpurun_out/$tag.$1
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/$tag.$1 -o r -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --groups 1 --config ${RSIM_CONFIG:-lift} ${RSIM_BENCH_EXTRA} > gpurun_out/$tag.$1.log 2>&1
  python tools/pmc_sum.py gpurun_out/$tag.$1 'k_step<' | tee gpurun_out/$tag.$1.txt; [ -n "$KEEP" ] || rm -rf gpurun_out/$tag.$1
}
for s in $sets; do
case $s in
 sq1) run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU";;
 sq2) run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA";;
 ic) run ic "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_BRANCH SQ_BUSY_CYCLES";;
 ld) run ld "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL";;
 vm) run vm "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES";;
 gr) run gr "GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVE_CYCLES";;
 hbm1) run hbm1 "FETCH_SIZE";;
 hbm2) run hbm2 "WRITE_SIZE";;
 hbm3) run hbm3 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum";;
 lv) run lv "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE";;
esac
done
