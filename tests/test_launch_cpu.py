"""lightning_asr_amd/launch.py on the CPU: the plain command starts its own ranks (as Lightning's DDP plugin does for the reference,
/root/reference/train.py:233-252 + conf/conf.yaml:21,30), the ranks under torch.distributed.run supervise a worker each, and a
failed attempt is retried in fresh processes one rung down the ladder.  The worker is tests/helpers/launch_worker.py (gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "launch_worker.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if not k.startswith(("LASR_", "TORCHELASTIC_")) and k not in
           ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def _run(cmd, env, timeout=240):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    js = None
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            js = json.loads(line)
    return r, js


def test_plain_command_starts_its_own_ranks():
    r, js = _run([sys.executable, WORKER, "2"], _env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert js["world"] == 2 and js["sum"] == 3.0
    assert js["launcher"]["rung_index"] == 0 and js["launcher"]["rung"] == "graph + lasr_comm"
    assert js["launcher"]["attempts"][0]["exit_codes"] == [0, 0]
    assert "some chatter" in r.stderr and "some chatter" not in r.stdout      # everything else rank 0 printed goes to stderr
    assert r.stdout.count('"metric"') == 1 and len(r.stdout.strip().splitlines()) == 1      # stdout is the ONE JSON line


def test_failed_rung_is_retried_in_fresh_processes_one_rung_down():
    r, js = _run([sys.executable, WORKER, "2"], _env(LASR_LAUNCH_FAULT="0:1"))
    assert r.returncode == 0, r.stderr[-2000:]
    la = js["launcher"]
    assert la["rung_index"] == 1 and la["rung"] == "eager + lasr_comm" and js["graph_dp"] == "0" and js["comm"] == "rccl"
    assert "rank 1 exited with code 7" in la["attempts"][0]["failed"] and la["attempts"][1]["failed"] is None
    # two rungs down: torch.distributed carries the gradients
    r, js = _run([sys.executable, WORKER, "2"], _env(LASR_LAUNCH_FAULT="0:0", LAUNCH_TEST_HANG="1:1", LASR_LAUNCH_TIMEOUT_S="8"))
    assert r.returncode == 0, r.stderr[-2000:]
    la = js["launcher"]
    assert la["rung_index"] == 2 and js["comm"] == "torch" and js["graph_dp"] == "0"
    assert "no result after" in la["attempts"][1]["failed"]            # the hung rung was ended by the supervisor's clock


def test_a_rank_that_hangs_while_another_dies_is_ended():
    r, js = _run([sys.executable, WORKER, "2"], _env(LASR_LAUNCH_FAULT="0:1", LAUNCH_TEST_HANG="0:0"))
    assert r.returncode == 0 and js["launcher"]["rung_index"] == 1
    assert js["launcher"]["attempts"][0]["seconds"] < 60


def test_switches_set_by_hand_collapse_the_ladder_and_exhaustion_is_an_error():
    r, js = _run([sys.executable, WORKER, "2"], _env(LASR_GRAPH_DP="0", LASR_LAUNCH_FAULT="0:1"))
    assert r.returncode == 0 and js["launcher"]["rung"] == "eager + torch.distributed" and js["launcher"]["rung_index"] == 1
    r, js = _run([sys.executable, WORKER, "2"], _env(LASR_LAUNCH_FAULT="0:1", LASR_LAUNCH_MAX_RUNGS="1"))
    assert r.returncode == 7 and js is None                 # no JSON line from a failed run, the worker's code comes back
    assert "rung 0 (graph + lasr_comm) failed" in r.stderr


@pytest.mark.parametrize("fault", [None, "0:1"])
def test_ranks_under_torch_distributed_run_supervise_a_worker_each(fault):
    from lightning_asr_amd.launch import free_port
    kw = {"LASR_LAUNCH_FAULT": fault} if fault else {}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), WORKER, "2"]
    r, js = _run(cmd, _env(**kw))
    assert r.returncode == 0, r.stderr[-3000:]
    assert js["world"] == 2 and js["sum"] == 3.0 and r.stdout.count('"metric"') == 1
    la = js["launcher"]
    assert "torch.distributed.run" in la["mode"] and la["rung_index"] == (1 if fault else 0)
    if fault:
        assert la["attempts"][0]["failed"] and js["graph_dp"] == "0"


def test_one_rank_and_hand_set_rank_are_workers():
    from lightning_asr_amd import launch
    old = dict(os.environ)
    try:
        for k in ("RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID", "LASR_LAUNCH_WORKER", "LASR_LAUNCH"):
            os.environ.pop(k, None)
        assert launch.role(1) == "worker" and launch.role(8) == "parent"
        os.environ.update(RANK="3", WORLD_SIZE="8")
        assert launch.role(8) == "worker"                    # RANK set by hand (or by another launcher): do as told
        os.environ["TORCHELASTIC_RUN_ID"] = "x"
        assert launch.role(8) == "rank_supervisor"
        os.environ["WORLD_SIZE"] = "1"
        assert launch.role(1) == "worker"
        os.environ.update(WORLD_SIZE="8", LASR_LAUNCH_WORKER="1")
        assert launch.role(8) == "worker"
    finally:
        os.environ.clear()
        os.environ.update(old)
