"""Which side moves when tests/test_gpu_model.py's golden case fails after tests/test_gpu_units.py ran in the same process:
the CPU oracle's f32 result as a function of torch's thread count, against one GPU result."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_model import golden_inputs, _native, rel_l2, R
from lightning_asr_amd import ops
dev = torch.device("cuda:0")
x, tg, pct, tsz = golden_inputs()
m = _native("plain", 28, dev)
feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
base = None
for n in (torch.get_num_threads(), 1, 2, 4, 8, 16, 32):
    torch.set_num_threads(n)
    o = R.OracleModel("plain", 28, mask=True, state=R.formula_state("plain", 28))
    st = R.NovogradState(len(o.parameters()))
    _, grads = R.train_step(o, st, x, tg, pct, tsz, 1e-2, 1e-3)
    rels = {t.name: rel_l2(m.view(t, m.grads), g) for t, g in zip(m.param_infos(), grads)}
    w = max(rels, key=rels.get)
    if base is None: base = [g.clone() for g in grads]
    d = max(((a - b).norm() / b.norm()).item() for a, b in zip(grads, base))
    print("threads %3d: gpu-vs-oracle worst %.3e (%s)   oracle-vs-oracle(default threads) worst %.3e" % (n, rels[w], w, d), flush=True)
