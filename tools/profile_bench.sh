#!/bin/bash
# rocprofv3 kernel-trace/stats of the default bench workload (run on the GPU box); summaries land in gpurun_out/
TAG=${1:-r01_c}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --headline-only > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -20
