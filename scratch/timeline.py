"""Ordered per-launch timeline of the last graph replay in a rocprofv3 --kernel-trace csv (after the last spin kernel):
start offset, duration, gap to the previous kernel's end, grid, workgroup, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_spin = max(i for i, r in enumerate(rows) if "spin" in r["Kernel_Name"].lower() or "sleep" in r["Kernel_Name"].lower())
rows = rows[last_spin + 1:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
tot = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]); wg = int(r["Workgroup_Size_X"])
    print("%9.1f  dur %7.1f  gap %6.1f  blocks %6d x %4d  lds %6s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g // max(wg, 1), wg, r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:70]))
    prev_end = max(prev_end, e); tot += (e - s) / 1e3
print("launches %d  kernel time %.1f us  span %.1f us" % (len(rows), tot, (prev_end - t0) / 1e3))
