"""Dev tool: summary of the plan's roctx ranges in a rocprofv3 (rocpd) database.
usage: python tools/roctx_summary.py <results.db> <out.txt>   (run: LASR_ROCTX=1 LASR_BENCH_GRAPH=0 rocprofv3 --marker-trace --kernel-trace -- python3 bench.py ...)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
names = {}
if "region_args" in tabs:
    cols = [r[1] for r in cur.execute("pragma table_info(region_args)")]
    idc = "id" if "id" in cols else cols[0]
    valc = "value" if "value" in cols else cols[-1]
    for rid, val in cur.execute("select %s, %s from region_args" % (idc, valc)):
        if isinstance(val, str) and val.startswith("lasr:"):
            names[rid] = val
rows = cur.execute("select id, name, start, end, extdata from regions").fetchall()
agg = {}
for rid, name, start, end, ext in rows:
    label = names.get(rid)
    if label is None and isinstance(ext, str) and "lasr:" in ext:
        i = ext.index("lasr:")
        label = ext[i:].split('"')[0]
    label = label or name
    a = agg.setdefault(label, [0, 0.0])
    a[0] += 1
    a[1] += (end - start) / 1000.0
out = ["host-side roctx ranges of the plan (csrc/model.hip, capi.hip; step.py adds lasr:step): name, count, mean HOST duration (us) - the",
       "time the host spent enqueueing the range's launches, not the kernels' time (join with the kernel trace by timestamp for that)", ""]
for label, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.append("%-44s %6d  %10.1f us" % (label, n, t / n))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
