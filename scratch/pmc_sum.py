"""Sum rocprofv3 --pmc counter csv rows per (kernel, counter); prints per-launch averages."""
import csv, sys, collections, glob, os
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (kn, cn), (n, v) in sorted(agg.items()):
    if os.environ.get("PMC_FILTER", "attn") in kn:
        print("%-40s %-32s launches %3d  per launch %.4g" % (kn, cn, n, v / n))
