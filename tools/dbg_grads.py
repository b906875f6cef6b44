import sys; sys.path.insert(0,'.')
import torch, numpy as np
from oracle import ref_cpu as R
from oracle.make_golden import golden_inputs
from lightning_asr_amd import ops
from lightning_asr_amd.engine import NativeModel
dev=torch.device('cuda')
x, tg, pct, tsz = golden_inputs()
m = NativeModel("plain", 28, True, "relu", torch.float32, device=dev); m.load_state_dict(R.formula_state("plain",28))
feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
loss, nll, lp, am = m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
o = R.OracleModel("plain", 28, mask=True, state=R.formula_state("plain", 28))
st = R.NovogradState(len(o.parameters()))
_, grads = R.train_step(o, st, x, tg, pct, tsz, 1e-2, 1e-3)
def rel(a,b): a,b=a.double().cpu(),b.double().cpu(); return ((a-b).norm()/(b.norm()+1e-30)).item()
for t,g in zip(m.param_infos(), grads):
    r=rel(m.view(t,m.grads),g)
    if r>5e-4: print(t.name, tuple(t.shape), r)
