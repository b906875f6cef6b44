"""Dev tool: how busy is the GPU inside a replayed step?  Reads a `rocprofv3 --kernel-trace --output-format csv` directory of
`bench.py` (graph replay), cuts the trace into steps at `novograd_update_kernel`, and reports per step: wall time (first start ->
last end), summed kernel time, idle time between kernels, and the kernel pairs with the largest idle time in front of the second.

usage: python tools/graph_gaps.py <trace dir> [steps to skip at the start]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "novograd_update_kernel" in r[2]:
            steps.append(cur)
            cur = []
    steps = steps[skip:]
    if not steps:
        print("no steps found")
        return
    walls, busys, idles = [], [], []
    pair_idle = defaultdict(lambda: [0.0, 0])
    for st in steps:
        t_end = st[0][0]
        busy = idle = 0.0
        for i, (s, e, n) in enumerate(st):
            if i and s > t_end:
                g = (s - t_end) / 1e3
                idle += g
                k = (st[i - 1][2].split("(")[0][-48:], n.split("(")[0][-48:])
                pair_idle[k][0] += g
                pair_idle[k][1] += 1
            busy += (e - max(s, t_end if i else s)) / 1e3 if e > t_end or not i else 0.0
            t_end = max(t_end, e)
        walls.append((t_end - st[0][0]) / 1e3)
        busys.append(busy)
        idles.append(idle)
    n = len(steps)
    print("steps %d  kernels/step %.1f" % (n, sum(len(s) for s in steps) / n))
    print("wall %.1f us  busy %.1f us  idle between kernels %.1f us (%.1f %%)" % (sum(walls) / n, sum(busys) / n, sum(idles) / n,
                                                                              100.0 * sum(idles) / sum(walls)))
    # step-to-step distance (includes the gap between replays)
    d2 = [(steps[i + 1][0][0] - steps[i][0][0]) / 1e3 for i in range(n - 1)]
    if d2:
        print("start-to-start %.1f us" % (sum(d2) / len(d2)))
    print("largest idle sites (us per step, count per step, us each):")
    for k, (g, c) in sorted(pair_idle.items(), key=lambda kv: -kv[1][0])[:15]:
        print("  %7.2f  %5.1f  %5.2f   %s  ->  %s" % (g / n, c / n, g / c, k[0], k[1]))


if __name__ == "__main__":
    main()
