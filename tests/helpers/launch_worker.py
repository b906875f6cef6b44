"""Stand-in entry point for tests/test_launch_cpu.py: goes through lightning_asr_amd.launch exactly as bench.py does, then - as a
worker - forms a gloo group from the environment the launcher gave it, all-reduces its rank and prints ONE JSON line on rank 0.
LAUNCH_TEST_HANG="rung:rank": that worker never returns (a rank stuck in a collective)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lightning_asr_amd import launch  # noqa: E402


def main():
    n = int(sys.argv[1])
    rc = launch.maybe_launch(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("LAUNCH_TEST_HANG") == "%s:%d" % (os.environ.get("LASR_LAUNCH_RUNG", "0"), rank):
        time.sleep(3600)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("some chatter before the result")
        print(json.dumps({"metric": "launch-test", "world": world, "sum": float(t.item()),
                          "graph_dp": os.environ.get("LASR_GRAPH_DP", "1"), "comm": os.environ.get("LASR_COMM", "rccl"),
                          "rung": launch.rung_info()}), flush=True)


if __name__ == "__main__":
    main()
