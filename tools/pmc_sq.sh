#!/bin/bash
# SQ counter passes of a solver workload (tools/pmc_rb.py): where a wave's cycles go.
# usage: tools/pmc_sq.sh TAG   (env: PDEIP_* knobs are inherited)
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/pmc_rb.py > $OUT/run.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $OUT
  i=$((i+1))
done
