"""In-situ report of the last graph replay in a rocprofv3 kernel trace csv (after the last torch elementwise marker kernel):
per kernel symbol the launches and the SUM of durations inside the two-stream step, the wall span, and how much of the span
has 0 / 1 / 2+ kernels in flight."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "elementwise" in r["Kernel_Name"])
rows = rows[last + 1:]
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0])
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl = 0; prev = t0; occ = collections.Counter()
for tm, d in ev:
    occ[min(lvl, 3)] += tm - prev; prev = tm; lvl += d
span = (t1 - t0) / 1e3
tot = sum(v[1] for v in agg.values())
print("launches %d  span %.1f us  sum of kernel durations %.1f us  (x%.2f)" % (len(rows), span, tot, tot / span))
print("time with k kernels in flight: " + "  ".join("%d: %.0f us (%.0f%%)" % (k, occ[k] / 1e3, 100.0 * occ[k] / 1e3 / span) for k in sorted(occ)))
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %5d launches %9.1f us  avg %7.1f  %5.1f%%" % (k, n, d, d / n, 100.0 * d / tot))
