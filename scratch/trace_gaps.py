"""GPU idle time inside graph replay from a rocprofv3 kernel trace: union of kernel intervals, gaps between them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
# keep the last image: kernels after the last gap > 5 ms that is followed by > 10000 kernels ... simpler: the densest window
busy = 0; gaps = []; cur_s, cur_e = iv[0][0], iv[0][1]
for s, e, n, q in iv[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
small = [g for g, _ in gaps if g < 100000]
print("kernels %d, busy (union) %.1f ms, gaps<100us: %d totalling %.1f ms (%.1f%% of busy), mean gap %.2f us" % (
    len(iv), busy / 1e6, len(small), sum(small) / 1e6, 100.0 * sum(small) / busy, sum(small) / max(1, len(small)) / 1e3))
h = collections.Counter(min(int(g / 1000), 20) for g in small)
print("gap histogram (us: count):", " ".join("%d:%d" % (k, h[k]) for k in sorted(h)))
tot = sum(e - s for s, e, _, _ in iv)
print("sum of kernel durations %.1f ms -> average concurrency %.2f" % (tot / 1e6, tot / busy))
after = collections.Counter()
for g, n in gaps:
    if 3000 < g < 100000: after[n.split("(")[0][:50]] += g
print("kernels that follow gaps of 3-100 us (total us):", [(k, round(v / 1e3)) for k, v in after.most_common(8)])
