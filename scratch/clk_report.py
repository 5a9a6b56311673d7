import csv, glob, sys, collections
d = sys.argv[1]
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in dur:
    t = dur[k][-1]
    print(k, "dur %.1f us" % (t / 1e3), {c: "%.4g -> %.2f GHz-equiv" % (v[-1], v[-1] / t) for c, v in cnt[k].items()})
