#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) of the red-black elin4 sweeps at 4K.
TAG=${1:-r01_d}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$ctr
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/pmc_rb.py > $OUT/run.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_FETCH_SIZE $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_WRITE_SIZE
