#!/bin/bash
# rocprofv3 kernel trace of the exact-order elin4 call at 4K (tools/time_exact4k.py): per-kernel average durations, for the
# walker and for round 2's kernel in the same run.  Usage (GPU box): bash tools/trace_exact.sh TAG [iters...]
TAG=${1:-x}; shift
ITERS=${@:-4}
cd /tmp && export TMPDIR=/tmp
for mode in walk old; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_${TAG}_$mode
  rm -rf $OUT; mkdir -p $OUT
  if [ $mode = old ]; then export PDEIP_EXACT_WALK=0; else export PDEIP_EXACT_WALK=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_exact4k.py $ITERS > $OUT/run.log 2>&1
  echo "== $mode"; grep iter $OUT/run.log
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v2 = sorted(v)
    print("  %-70s n=%4d  mean %8.1f us  median %8.1f  min %8.1f" % (n[-70:], len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
PY
done
