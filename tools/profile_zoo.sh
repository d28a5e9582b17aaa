#!/bin/bash
# rocprofv3 passes over tools/profile_zoo.py: kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in their own runs.
TAG=${1:-r02_c}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/zoo_${TAG}_stats; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/profile_zoo.py > $OUT/run.log 2>&1 || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  OUT=$R/gpurun_out/zoo_${TAG}_$ctr; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/profile_zoo.py > $OUT/run.log 2>&1 || exit 1
done
python3 $R/tools/summarize_zoo.py $TAG
