import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_sor" in r["Kernel_Name"] or "k_derive" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print("%-62s %-12s n=%3d mean=%.4g  min=%.4g max=%.4g" % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
