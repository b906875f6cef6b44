"""Dev experiment: does computing the NEXT batch's features on a second stream while the current step runs hide
the mel kernel (and by how much)?  Same work per step either way."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lightning_asr_amd.engine import NativeModel
from lightning_asr_amd.step import TrainStep
dev = torch.device("cuda")
model = NativeModel("plain", bench.V + 1, mask=True, act="relu", dtype=torch.bfloat16, device=dev)
model.init_parameters(seed=0)
ts = TrainStep(model, 1e-2, 1e-3)
wave, tg, tl = bench.synth_batch(32, int(bench.CLIP_S * bench.SR), bench.S_TGT, 1234, dev)
def run_plain(n):
    for _ in range(n): ts.step(wave, tg, tl)
prio = int(os.environ.get("PRIO", "0"))
side = torch.cuda.Stream(priority=prio)
def run_pref(n):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        nxt = ts.features(wave)
    for _ in range(n):
        main.wait_stream(side)
        feats, pct = nxt
        with torch.cuda.stream(side):
            nxt = ts.features(wave)         # next batch's features: overlaps with this step
        ts.step_features(feats, pct, tg, tl)
for name, fn in (("plain", run_plain), ("prefetch", run_pref), ("plain", run_plain), ("prefetch", run_pref)):
    fn(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(40); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print("%-9s %.3f ms/step  %.0f audio-s/s" % (name, dt * 1e3, 320 / dt))
