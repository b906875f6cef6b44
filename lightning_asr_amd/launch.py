"""Who starts the per-GPU ranks, and what happens when the first attempt does not survive.

The reference gets its data-parallel ranks from Lightning: ``python train.py`` with ``gpus: N`` + ``accelerator: ddp``
(`/root/reference/conf/conf.yaml:21,30`, `train.py:233-252`) - the DDP plugin of Lightning 1.3 starts the per-GPU children itself
from the plain command.  This module gives ``python bench.py --gpus N`` and ``python -m lightning_asr_amd.train train.gpus=N`` the
same property, and puts a fallback ladder around the ranks:

    rung 0   the defaults: the staged step with its RCCL all-reduces captured in a hipGraph, `lasr_comm_*` on the side stream
    rung 1   LASR_GRAPH_DP=0          eager launches, `lasr_comm_*` still carries the gradients
    rung 2   + LASR_COMM=torch        eager launches, torch.distributed's all-reduce (nccl backend = RCCL)

Every rung runs in FRESH worker processes (a rank that hung in a collective or lost its HIP context is not reused); the supervisor
never initialises the GPU and never execs - it only starts children (`subprocess.Popen`), relays rank 0's output and exits with the
workers' code.  Three ways in:

    plain command, N > 1, no RANK in the environment     -> this process is the PARENT: it starts N workers (RANK / LOCAL_RANK /
                                                            WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT per rung)
    under `python -m torch.distributed.run` (N > 1)      -> every rank is a RANK SUPERVISOR of one worker; the supervisors agree
                                                            on "rung failed" through the agent's TCPStore (no GPU, no collective)
    RANK set by hand / LASR_LAUNCH_WORKER=1 / N = 1      -> this process IS the worker

Pure stdlib + torch.distributed.TCPStore (CPU); importing this module touches no device.
"""
from __future__ import annotations

import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple

LADDER: Tuple[Tuple[str, Dict[str, str]], ...] = (
    ("graph + lasr_comm", {}),
    ("eager + lasr_comm", {"LASR_GRAPH_DP": "0"}),
    ("eager + torch.distributed", {"LASR_GRAPH_DP": "0", "LASR_COMM": "torch"}),
)


def role(n_ranks: int) -> str:
    """'worker' | 'parent' | 'rank_supervisor' for this process (see the module docstring)"""
    env = os.environ
    if env.get("LASR_LAUNCH_WORKER") == "1" or env.get("LASR_LAUNCH") == "0":
        return "worker"
    if "RANK" in env:
        if int(env.get("WORLD_SIZE", "1")) > 1 and "TORCHELASTIC_RUN_ID" in env:
            return "rank_supervisor"
        return "worker"
    return "parent" if n_ranks > 1 else "worker"


_SWITCHES = ("LASR_GRAPH_DP", "LASR_COMM")


def rungs() -> List[Tuple[str, Dict[str, str]]]:
    """the ladder as (name, environment to add); a switch the caller already set wins, and rungs that come out equal collapse
    (LASR_GRAPH_DP=0 set by hand: two rungs, both eager)"""
    out, seen = [], set()
    for _name, extra in LADDER:
        eff = tuple(os.environ.get(k, extra.get(k, "")) for k in _SWITCHES)
        if eff in seen:
            continue
        seen.add(eff)
        name = ("eager" if eff[0] == "0" else "graph") + " + " + ("torch.distributed" if eff[1] == "torch" else "lasr_comm")
        out.append((name, {k: v for k, v in extra.items() if k not in os.environ}))
    limit = int(os.environ.get("LASR_LAUNCH_MAX_RUNGS", str(len(out))))
    return out[:max(1, limit)]


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Worker:
    """one child process; rank 0's stdout is read line by line (the bench's JSON line is held back until the rung is known good)"""

    def __init__(self, cmd: Sequence[str], env: Dict[str, str], capture: bool, stream: bool):
        self.lines: List[str] = []
        self.stream = stream
        # rank 0: stdout is read here; every other rank's stdout goes to OUR stderr - stdout of the whole run carries what rank 0
        # printed and nothing else (gloo, for one, announces its peers on stdout from every rank)
        self.p = subprocess.Popen(list(cmd), env=env, stdout=subprocess.PIPE if capture else sys.stderr, text=capture,
                                  bufsize=1 if capture else -1)
        self.t = None
        if capture:
            self.t = threading.Thread(target=self._pump, daemon=True)
            self.t.start()

    def _pump(self):
        for line in self.p.stdout:
            self.lines.append(line)
            if self.stream:
                sys.stdout.write(line)
                sys.stdout.flush()

    def stop(self):
        """end exactly this child (never a pattern): SIGTERM, ten seconds, SIGKILL"""
        if self.p.poll() is None:
            self.p.terminate()
            try:
                self.p.wait(10)
            except subprocess.TimeoutExpired:
                self.p.kill()
                self.p.wait()
        if self.t is not None:
            self.t.join(5)


_LIVE: List[_Worker] = []


def _on_signal(signum, _frame):
    for w in list(_LIVE):
        w.stop()
    sys.exit(128 + signum)


def _install_signals():
    for s in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(s, _on_signal)
        except ValueError:          # not the main thread (tests)
            pass


def _timeout_s() -> float:
    return float(os.environ.get("LASR_LAUNCH_TIMEOUT_S", "900"))


def _worker_env(rung: int, extra: Dict[str, str], more: Dict[str, str]) -> Dict[str, str]:
    env = dict(os.environ)
    env.update(extra)
    env.update(more)
    env["LASR_LAUNCH_WORKER"] = "1"
    env["LASR_LAUNCH_RUNG"] = str(rung)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    return env


def _last_json(lines: List[str]) -> Optional[dict]:
    for line in reversed(lines):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def _emit(lines: List[str], record: dict, hold_json: bool) -> None:
    """hold_json mode: rank 0's last JSON line, with the `launcher` record merged in, is the ONLY thing written to stdout"""
    if not hold_json:
        return                               # (streamed live)
    js = _last_json(lines)
    for line in lines:                       # whatever else rank 0 said goes to stderr: stdout is the ONE JSON line
        s = line.strip()
        if js is not None and s.startswith("{") and s.endswith("}"):
            continue
        sys.stderr.write(line)
    sys.stderr.flush()
    if js is not None:
        js["launcher"] = record
        sys.stdout.write(json.dumps(js) + "\n")
    sys.stdout.flush()


def run_parent(cmd: Sequence[str], n: int, hold_json: bool = True) -> int:
    """plain command, N > 1: start N workers per rung, supervise, relay rank 0; returns the exit code"""
    _install_signals()
    attempts = []
    ladder = rungs()
    last_rc = 1
    for r, (name, extra) in enumerate(ladder):
        port = free_port()
        t0 = time.time()
        ws: List[_Worker] = []
        for k in range(n):
            env = _worker_env(r, extra, {"RANK": str(k), "LOCAL_RANK": str(k), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                                         "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
            w = _Worker(cmd, env, capture=(k == 0), stream=not hold_json)
            ws.append(w)
            _LIVE.append(w)
        rcs: List[Optional[int]] = [None] * n
        why = None
        while True:
            for k, w in enumerate(ws):
                if rcs[k] is None:
                    rcs[k] = w.p.poll()
            if any(rc not in (None, 0) for rc in rcs):
                bad = next((k, rc) for k, rc in enumerate(rcs) if rc not in (None, 0))
                why, last_rc = "rank %d exited with code %d" % bad, bad[1]
                break
            if all(rc == 0 for rc in rcs):
                break
            if time.time() - t0 > _timeout_s():
                why = "no result after %.0f s (LASR_LAUNCH_TIMEOUT_S)" % _timeout_s()
                break
            time.sleep(0.1)
        for w in ws:
            w.stop()
            _LIVE.remove(w)
        attempts.append({"rung": name, "env": extra, "exit_codes": [w.p.returncode for w in ws], "seconds": round(time.time() - t0, 1),
                         "failed": why})
        if why is None:
            _emit(ws[0].lines, {"mode": "plain command: %d fresh worker processes started by the parent (no outer launcher)" % n,
                                "rung": name, "rung_index": r, "attempts": attempts}, hold_json)
            return 0
        sys.stderr.write("[lasr launch] rung %d (%s) failed: %s%s\n" % (r, name, why, "; next rung in fresh processes" if r + 1 < len(ladder) else ""))
        if hold_json:
            for line in ws[0].lines:         # whatever rank 0 said before it fell: to stderr, never mistaken for the result
                sys.stderr.write("[rank 0, rung %d] %s" % (r, line))
    return last_rc if 0 < last_rc < 256 else 1       # the code of the worker that fell first on the last rung (1: killed / timed out)


def run_rank_supervisor(cmd: Sequence[str], hold_json: bool = True) -> int:
    """under torch.distributed.run: this rank supervises ONE worker per rung; 'somebody failed' travels through the agent's TCPStore"""
    from datetime import timedelta
    from torch.distributed import TCPStore
    _install_signals()
    env0 = os.environ
    rank, world = int(env0["RANK"]), int(env0["WORLD_SIZE"])
    host, port = env0.get("MASTER_ADDR", "127.0.0.1"), int(env0["MASTER_PORT"])
    pre = "lasr_launch/%s/%s/" % (env0.get("TORCHELASTIC_RUN_ID", "-"), env0.get("TORCHELASTIC_RESTART_COUNT", "0"))
    try:
        store = TCPStore(host, port, None, False, timeout=timedelta(seconds=60))
    except Exception as e:                    # no agent store to talk through: be the worker, as before this module existed
        sys.stderr.write("[lasr launch] no rendez-vous store at %s:%d (%s): running unsupervised\n" % (host, port, e))
        return -1
    attempts = []
    ladder = rungs()
    for r, (name, extra) in enumerate(ladder):
        more: Dict[str, str] = {}
        if r > 0:
            # the first attempt's keys are still in the agent's store: later rungs meet on a store of their own, hosted by worker 0
            if rank == 0:
                store.set(pre + "port%d" % r, str(free_port()))
            store.wait([pre + "port%d" % r], timedelta(seconds=120))
            more = {"MASTER_PORT": store.get(pre + "port%d" % r).decode(), "TORCHELASTIC_USE_AGENT_STORE": "False"}
        fail_key, done_key = pre + "fail%d" % r, pre + "done%d" % r
        t0 = time.time()
        w = _Worker(cmd, _worker_env(r, extra, more), capture=(rank == 0), stream=not hold_json)
        _LIVE.append(w)
        why = None
        while True:
            rc = w.p.poll()
            if rc is not None:
                if rc != 0:
                    why = "rank %d exited with code %d" % (rank, rc)
                break
            if store.check([fail_key]):
                why = "another rank failed (%s)" % store.get(fail_key).decode()
                break
            if time.time() - t0 > _timeout_s():
                why = "rank %d: no result after %.0f s (LASR_LAUNCH_TIMEOUT_S)" % (rank, _timeout_s())
                break
            time.sleep(0.1)
        w.stop()
        _LIVE.remove(w)
        if why is not None and not store.check([fail_key]):
            store.set(fail_key, why)
        store.add(done_key, 1)
        t1 = time.time()
        while store.add(done_key, 0) < world and time.time() - t1 < 120:
            time.sleep(0.05)
        failed = store.check([fail_key])
        attempts.append({"rung": name, "env": extra, "exit_code_rank0": w.p.returncode, "seconds": round(time.time() - t0, 1),
                         "failed": store.get(fail_key).decode() if failed else None})
        if not failed:
            if rank == 0:
                _emit(w.lines, {"mode": "torch.distributed.run: every rank supervises a fresh worker process per rung", "rung": name,
                                "rung_index": r, "attempts": attempts}, hold_json)
            return 0
        if rank == 0:
            sys.stderr.write("[lasr launch] rung %d (%s) failed: %s%s\n" % (r, name, attempts[-1]["failed"],
                                                                            "; next rung in fresh processes" if r + 1 < len(ladder) else ""))
            if hold_json:
                for line in w.lines:
                    sys.stderr.write("[rank 0, rung %d] %s" % (r, line))
    rc = w.p.returncode
    return rc if rc is not None and 0 < rc < 256 else 1


def maybe_launch(n_ranks: int, cmd: Sequence[str], hold_json: bool = True) -> Optional[int]:
    """Call FIRST in an entry point, before anything touches the GPU.  Returns None when this process is a worker (carry on), else
    the exit code of the supervised run (the caller exits with it)."""
    who = role(n_ranks)
    if who == "worker":
        fault = os.environ.get("LASR_LAUNCH_FAULT")          # tests: "rung:rank" makes that worker of that rung die at once
        if fault and fault == "%s:%s" % (os.environ.get("LASR_LAUNCH_RUNG", "0"), os.environ.get("RANK", "0")):
            sys.stderr.write("[lasr launch] injected fault (LASR_LAUNCH_FAULT=%s)\n" % fault)
            sys.exit(7)
        return None
    if who == "parent":
        return run_parent(cmd, n_ranks, hold_json)
    rc = run_rank_supervisor(cmd, hold_json)
    return None if rc == -1 else rc


def rung_info() -> Optional[dict]:
    """for a worker's own record: which rung of the ladder it runs on (None: not started through this module)"""
    if os.environ.get("LASR_LAUNCH_WORKER") != "1":
        return None
    r = int(os.environ.get("LASR_LAUNCH_RUNG", "0"))
    return {"rung_index": r}
