"""Dev tool (GPU box): sustained manifest -> GPU rate of the ingest alone (wav decode by liblasr's host threads into the pinned ring,
int16 H2D on the copy stream, log-mel features on the device; no model) as a function of the reader thread count - what the
reference does with `num_worker` DataLoader processes (data_module.py:199-201, conf/conf.yaml:14).
usage: python tools/ingest_rate.py [--threads 1,2,4,8,16] [--batches 200]"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lightning_asr_amd import ops  # noqa: E402
from lightning_asr_amd.data_module import AudioParser, MyAudioDataset  # noqa: E402
from lightning_asr_amd.ingest import BatchProducer, DeviceFeeder, PinnedRing, SR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,2,4,8,16")
    ap.add_argument("--batches", type=int, default=200)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    labels = [c.strip() for c in open(os.path.join(ROOT, "data", "labels.txt"), encoding="utf-8").readlines()]
    root = tempfile.mkdtemp(prefix="lasr_ingest_")
    cfg = bench.CONFIGS["cfg2"]
    man, _ = bench.write_corpus(root, cfg, labels, a.batch, 8, a.batches, 4321)
    ds = MyAudioDataset([man], labels, mask=True)
    parser = AudioParser(device=str(dev))
    out = {}
    for nt in [int(x) for x in a.threads.split(",")]:
        idx = [list(range(i * a.batch, (i + 1) * a.batch)) for i in range(a.batches)]
        ring = PinnedRing(4, a.batch * (int(10.0 * SR) + 64), 8 * a.batch + 2 * a.batch * 256)
        feeder = DeviceFeeder(ring, dev, n_slots=4)
        prod = BatchProducer(ds, idx, ring, True, parser, n_threads=nt, feeder=feeder)
        dd = ops.DeviceDither(1, dev)
        prod.start()
        secs, t0, n = 0.0, None, 0
        while True:
            db = prod.out.get()
            if db is None:
                break
            if isinstance(db, BaseException):
                raise db
            torch.cuda.current_stream().wait_event(db.ready)
            ops.mel(db.pcm, db.lens, dd, db.aug, True, torch.bfloat16, want_bft=False)
            feeder.release(db)
            n += 1
            if n == 20:                       # warm-up: page cache, allocator
                torch.cuda.synchronize()
                t0, secs = time.perf_counter(), 0.0
            secs += db.seconds
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        feeder.close()
        out[str(nt)] = {"audio_seconds_per_sec": secs / dt, "ms_per_batch": 1e3 * dt / (n - 20)}
        print("threads %2d: %.0f audio-s/s (%.2f ms per batch of %d x 10 s)" % (nt, secs / dt, 1e3 * dt / (n - 20), a.batch), flush=True)
    print(json.dumps({"ingest_rate": out, "what": "wav decode + crop + int16 H2D + log-mel (bf16) of 32 x 10 s batches, no model"}))
    import shutil
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
