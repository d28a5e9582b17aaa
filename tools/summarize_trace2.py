"""Kernel mix of a rocprofv3 kernel trace: per kernel calls / total / avg, plus wall span and the idle gaps between kernels."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    acc = collections.defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        k = n.replace("pdeip::", "").replace("void ", "").split("(")[0][:70]
        acc[k][0] += 1; acc[k][1] += e - s
    busy = sum(v[1] for v in acc.values()); span = rows[-1][1] - rows[0][0]
    gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    small = [g for g in gaps if 0 <= g < 100000]
    print("%s: %d kernels, busy %.2f ms, span %.2f ms, gaps<100us: n=%d sum %.2f ms median %.1f us" % (d, len(rows), busy / 1e6, span / 1e6, len(small), sum(small) / 1e6, sorted(small)[len(small) // 2] / 1e3))
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:22]:
        print("  %-72s %5d %9.3f ms  avg %7.1f us" % (k, n, t / 1e6, t / n / 1e3))
