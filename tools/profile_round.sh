#!/bin/bash
# One round's rocprofv3 evidence in one go (GPU box): usage tools/profile_round.sh TAG
#   1. bench headline under --kernel-trace --stats            -> gpurun_out/prof_TAG
#   2. every point-SOR kernel family at 4K: trace + FETCH_SIZE + WRITE_SIZE passes (separate runs) -> gpurun_out/zoo_TAG_*
#   3. SQ counter passes of the headline kernel (four separate runs) -> gpurun_out/pmc_TAG_sq_*.txt
#   4. kernel traces of the line-relaxation kernels, the small-frame solver and the stage kernels of a level / the FMG driver
TAG=${1:-r03_a}
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_bench.sh $TAG > $R/gpurun_out/prof_${TAG}.log 2>&1
bash $R/tools/profile_zoo.sh $TAG > $R/gpurun_out/zoo_${TAG}.log 2>&1
PDEIP_PERSIST_XCD=1 ZOO_TAG_SUFFIX=_xcd bash $R/tools/profile_zoo.sh ${TAG}_xcd > $R/gpurun_out/zoo_${TAG}_xcd.log 2>&1
bash $R/tools/pmc_sq.sh ${TAG}_sq > $R/gpurun_out/pmc_${TAG}_sq.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for what in "alr:time_alr.py 2160 3840 4" "small:time_small.py" "levels:levels_profile.py" "fmg:fmg_profile.py"; do
  name=${what%%:*}; cmd=${what#*:}
  OUT=$R/gpurun_out/trace_${TAG}_$name
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/$cmd > $OUT/run.log 2>&1
done
echo profiled
