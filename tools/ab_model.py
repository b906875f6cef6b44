"""A/B two builds of liblasr.so on the whole-model backward (dev tool)."""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import torch
from lightning_asr_amd import _lib, ops
from oracle import ref_cpu as R
from oracle.make_golden import golden_inputs
from tools.ab_ops import load

def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
dev = torch.device('cuda')
x, tg, pct, tsz = golden_inputs()
res = []
for path in ('build/liblasr_old.so', 'lightning_asr_amd/liblasr.so'):
    _lib._lib = load(os.path.abspath(path))
    from lightning_asr_amd.engine import NativeModel
    m = NativeModel("plain", 28, True, "relu", torch.float32, device=dev); m.load_state_dict(R.formula_state("plain", 28))
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    loss, nll, lp, am = m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    res.append((m, m.grads.clone(), lp.clone(), m.tap("grad_logits").clone()))
(m0, g0, lp0, gl0), (m1, g1, lp1, gl1) = res
print("logp", rel(lp1, lp0), "grad_logits", rel(gl1, gl0))
for t in m0.param_infos():
    r = rel(m1.view(t, g1), m0.view(t, g0))
    if r > 1e-5:
        print(t.name, "%.2e" % r)
