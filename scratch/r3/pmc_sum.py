"""Sum rocprofv3 --pmc counter csv rows per (kernel, grid, counter); prints per-launch averages.  PMC_FILTER: substrings
(comma separated) a kernel name must contain one of."""
import csv, sys, collections, os
agg = collections.defaultdict(lambda: [0, 0.0])
flt = [s for s in os.environ.get("PMC_FILTER", "").split(",") if s]
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0]
        if flt and not any(s in kn for s in flt):
            continue
        k = (kn[:48], r.get("Grid_Size", "?"), r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (kn, gs, cn), (n, v) in sorted(agg.items()):
    print("%-48s grid %-9s %-30s launches %3d  per launch %.5g" % (kn, gs, cn, n, v / n))
