"""Aggregate a rocprofv3 kernel-trace csv by kernel name, restricted to the launches after the last spin kernel."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
spin = [i for i, r in enumerate(rows) if "spin" in r["Kernel_Name"].lower()]
win = rows[spin[-1] + 1:] if spin else rows
agg = collections.OrderedDict()
for r in win:
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
print("kernels after the last spin: %d, total %.1f us" % (len(win), tot / 1e3))
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-90s %5d %9.1f us %7.2f avg" % (name[:90], a[0], a[1] / 1e3, a[1] / 1e3 / a[0]))
