#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  OUT=$R/gpurun_out/exact_pmc_$i; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/exact_pmc.py > $OUT/run.log 2>&1
  python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not f: print("no counters in", sys.argv[1]); sys.exit(0)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_sor_exact_persist" in r["Kernel_Name"]: acc[(r["Grid_Size"],r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, "n=%d mean=%.4g"%(len(v),sum(v)/len(v)))
PY
  i=$((i+1))
done
