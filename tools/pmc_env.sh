#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the red-black elin4 launch at 4K under an environment setting:  bash tools/pmc_env.sh TAG [VAR=value ...]
TAG=${1:-env}; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$ctr
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/pmc_rb.py > $OUT/run.log 2>&1
done
echo "== $TAG $*"
python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_FETCH_SIZE $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_WRITE_SIZE
