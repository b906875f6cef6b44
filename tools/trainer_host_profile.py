"""Dev tool (GPU box): cProfile of the main thread of `bench.py --path trainer` (where the host time of a fused step goes)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--path", "trainer", "--steps", "60", "--warmup", "10"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("cumulative")
    for pat in ("fused_fit.py:.*\\(step\\)", "step.py:.*step_features", "engine.py:.*loss_backward", "step.py:.*optimizer_step", "engine.py:.*arm_prefetch",
                "asr_metrics.py:.*device_distances", "fused_fit.py:.*run_epoch"):
        st.print_callees(pat)
