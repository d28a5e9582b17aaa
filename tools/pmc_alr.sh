#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the line-relaxation kernels at 2160x3840 (tools/time_alr.py 2160 3840 1): bash tools/pmc_alr.sh TAG
TAG=${1:-alr}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$ctr
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_alr.py 2160 3840 1 > $OUT/run.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_FETCH_SIZE $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_WRITE_SIZE
