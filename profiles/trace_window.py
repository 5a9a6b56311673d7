"""Per-kernel stats of bench.py's eager roofline pass from a rocprofv3 --kernel-trace csv.

    python3 profiles/trace_window.py <..._kernel_trace.csv> > profiles/r01_roofline_pass_kernel_stats.csv

bench.py's sampling loop replays a captured graph in which the two CFG halves are two concurrent batch-1 passes; its
roofline pass then runs ONE eager batch-2 UNet step and one decode, each behind a torch.cuda._sleep spin kernel (so
that the launches queue up).  Kernel names are the same in both, so the whole-run --stats average mixes batch-1
launches that overlap with another stream's kernels and batch-2 launches that run alone.  This script keeps only the
launches after the first spin kernel: the population bench.py's HIP-event figures (roofline.avg_launch_ms) refer to.
"""
import csv, sys, collections, re

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
spin = [i for i, r in enumerate(rows) if "spin" in r["Kernel_Name"].lower() or "sleep" in r["Kernel_Name"].lower()]
assert spin, "no spin kernel in the trace: was the roofline pass run?"
win = rows[spin[0] + 1:]
# the measured-peak probes run after the pass: not part of it
win = [r for r in win if "probe_" not in r["Kernel_Name"] and "spin" not in r["Kernel_Name"].lower() and "sleep" not in r["Kernel_Name"].lower()]
agg = collections.OrderedDict()
for r in win:
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([name, a[0], a[1], "%.1f" % (a[1] / a[0]), "%.2f" % (100.0 * a[1] / tot), a[2], a[3]])
