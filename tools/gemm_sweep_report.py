"""Kernel-only durations of tools/gemm_sweep.py from its rocprofv3 rocpd database."""
import sqlite3, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CONFIGS = [(n, k, tb) for tb in (0, 1) for n in (256, 512, 1024) for k in (64, 128, 256, 512, 1024)]
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, duration, grid_x, workgroup_x from kernels where name like '%gemm_bf16%' order by start"))
assert len(rows) == reps * len(CONFIGS), (len(rows), reps * len(CONFIGS))
for i, (N, K, tB) in enumerate(CONFIGS):
    seg = rows[i * reps:(i + 1) * reps]
    d = sorted(r[1] for r in seg[2:])
    fl = 2 * 2.0 * 16032 * N * K
    print("N=%4d K=%4d tB=%d %-8s tiles=%4d  median %6.1f us  min %6.1f  (%5.0f TFLOP/s)" % (
        N, K, tB, "big" if "big" in seg[0][0] else "small", seg[0][2] // seg[0][3], d[len(d) // 2] / 1e3, d[0] / 1e3, fl / d[len(d) // 2] / 1e3))
