#!/bin/bash
# PMC passes over the fused kernel (counters only: no trace domains besides kernel-trace).  Usage: tools/pmc_pass.sh <tag>
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { # name, counters
  rm -rf gpurun_out/$tag.$1
  timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/$tag.$1 -o r -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/$tag.$1.log 2>&1
  python tools/pmc_sum.py gpurun_out/$tag.$1 k_step | tee gpurun_out/$tag.$1.txt
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
