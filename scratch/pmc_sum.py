"""Sum rocprofv3 --pmc counters per kernel name from the csv output dir."""
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(r["Kernel_Name"][:40], r["Counter_Name"])] += 1
for k, v in acc.items():
    for c, x in sorted(v.items()):
        print(k, c, "%.4g" % (x / n[(k, c)]), "per launch x", n[(k, c)])
