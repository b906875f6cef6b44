"""Dev tool: from a rocprofv3 --kernel-trace CSV, report how much of each kernel's run time overlaps other kernels
(concurrency across streams).  python tools/overlap_report.py <kernel_trace.csv> [name-substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "dwconv_wgrad"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
tot = ov = n = 0
for i, (s, e, name, q) in enumerate(ev):
    if pat not in name:
        continue
    n += 1
    tot += e - s
    for j in range(max(0, i - 8), min(len(ev), i + 9)):
        if j == i:
            continue
        s2, e2 = ev[j][0], ev[j][1]
        ov += max(0, min(e, e2) - max(s, s2))
print("%d launches matching %r: %.1f us avg, %.1f%% of their run time overlapped by neighbours; queues seen: %s"
      % (n, pat, tot / max(n, 1) / 1e3, 100.0 * ov / max(tot, 1), sorted({e[3] for e in ev})))
