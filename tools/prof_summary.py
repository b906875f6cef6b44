"""Print the per-kernel summary of a rocprofv3 (rocpd sqlite) run: python tools/prof_summary.py results.db [steps] [out.csv]"""
import csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(top_kernels)")]
rows = list(c.execute("select * from top_kernels"))
arg = sys.argv[2] if len(sys.argv) > 2 else "1"
if arg.startswith("auto:"):        # steps = calls of a kernel that runs exactly once per step (counts graph-capture warm-ups too)
    hit = [r for r in rows if arg[5:] in r[0]]
    if not hit:
        sys.exit("no kernel matching %r in the trace" % arg[5:])
    steps = float(hit[0][1])
else:
    steps = float(arg)
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as f:
        w = csv.writer(f); w.writerow(cols); w.writerows(rows)
tot = sum(r[2] for r in rows)
print("total %.3f ms/step over %g steps" % (tot / steps / 1e3, steps))
for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 24]:
    print("%8.3f ms/step %6.1f calls/step %8.2f us avg %5.1f%%  %s" % (r[2] / steps / 1e3, r[1] / steps, r[3], r[4], r[0][:110]))
