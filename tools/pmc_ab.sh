#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the red-black elin4 launch at 4K for the library in tree and for a variant (tools/build_variant.py):
#   bash tools/pmc_ab.sh TAG [variant-name]
TAG=${1:-ab}
VAR=$2
cd /tmp && export TMPDIR=/tmp
for which in main $VAR; do
  if [ "$which" = main ]; then unset PDEIP_LIB; else export PDEIP_LIB=$GRAFT_REPO_ROOT/pde-based-image-processing_amd/libpdeip_$which.so; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${which}_$ctr
    rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/pmc_rb.py > $OUT/run.log 2>&1
  done
  echo "== $which"
  python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${which}_FETCH_SIZE $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${which}_WRITE_SIZE
done
