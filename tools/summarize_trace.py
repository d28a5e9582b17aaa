import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_sor_exact" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 3
rows = rows[2 * n:]  # last call
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
print("launches", len(d), "sum_us %.1f" % sum(d), "span_us %.1f" % ((int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3),
      "mean gap %.2f" % (sum(gaps) / len(gaps)))
for i in range(0, len(d), 6):
    print("m=%3d.." % i, " ".join("%5.1f" % x for x in d[i:i + 6]))
print("VGPR", rows[0].get("VGPR_Count"), "LDS", rows[0].get("LDS_Block_Size"), "grid", rows[0].get("Grid_Size"), rows[0].get("Workgroup_Size"))
