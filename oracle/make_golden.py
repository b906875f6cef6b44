"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference, dev
container only) on formula-defined weights/inputs, and check the oracle restatement
(oracle/ref_cpu.py) against it while doing so.  TEST INFRASTRUCTURE ONLY.

    python oracle/make_golden.py            # writes tests/golden/, asserts oracle == reference

Only OUTPUTS are stored; weights and inputs are regenerated from formulas on both sides
(ref_cpu.formula_state / golden_inputs).  The reference source never travels.
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

REF_MODULE = {"plain": "models.QuartNet", "context": "models.QuartNetContext",
              "context_se": "models.QuartNetContextSE"}


def golden_inputs(B: int = 4, T_in: int = 201, S: int = 12, V: int = 27):
    """Formula-defined batch: smooth-but-rich features, pct with a short and a ragged sample,
    LCG targets.  Regenerated identically by the tests."""
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    f = torch.arange(64, dtype=torch.float64).view(1, 1, 64, 1)
    t = torch.arange(T_in, dtype=torch.float64).view(1, 1, 1, T_in)
    x = torch.sin(0.013 * (t + 1) * (f + 3) + 0.7 * b) + 0.5 * torch.cos(0.31 * t + 0.9 * f - b) \
        + 0.25 * torch.sin(2.1 * t + 0.05 * f * f)
    pct = torch.tensor([1.0, 1.0, 0.9, 0.5], dtype=torch.float32)[:B]
    # frames past each sample's length are zero, as the collate would leave them
    lens_in = (T_in * pct).int()
    keep = (torch.arange(T_in).view(1, 1, 1, T_in) < lens_in.view(B, 1, 1, 1))
    x = (x * keep).float()
    state = 12345
    tg = torch.zeros(B, S, dtype=torch.int64)
    tsz = torch.tensor([S, S - 3, S - 5, S - 7], dtype=torch.int32)[:B]
    for i in range(B):
        for s in range(S):
            state = (1103515245 * state + 12345) % (1 << 31)
            tg[i, s] = (state >> 8) % V
        for s in range(int(tsz[i]), S):
            tg[i, s] = 0
    return x, tg, pct, tsz


def checksum(t: torch.Tensor) -> np.ndarray:
    """[mean, mean|.|, 8 samples] of a tensor — small, order-insensitive + positional."""
    f = t.detach().double().flatten()
    n = f.numel()
    idx = torch.linspace(0, n - 1, 8).long()
    return np.concatenate([[f.mean().item(), f.abs().mean().item()], f[idx].numpy()])


def swap_activation(model, act_mod):
    """The reference imports Swish / Mish (models/QuartNet.py:5) but constructs nn.ReLU everywhere.  To pin the
    oracle's ``act="swish"`` path, the unit epilogues of a reference model instance are replaced by the reference's OWN
    activate_fun.Swish.Swish module: SeprationConv.relu (models/QuartNet.py:25,37), QuartNetBlock.last_relu (:69,77) and the
    activation inside encoder.last_cnn2 (:148).  SELayer's internal ReLU (models/QuartNetContextSE.py:14) is part of the SE
    MLP, not an epilogue, and stays."""
    n = 0
    for name, sub in model.named_modules():
        for attr in ("relu", "last_relu"):
            if isinstance(getattr(sub, attr, None), torch.nn.ReLU):
                setattr(sub, attr, act_mod())
                n += 1
        if name.endswith("last_cnn2"):
            for i, child in enumerate(sub):
                if isinstance(child, torch.nn.ReLU):
                    sub[i] = act_mod()
                    n += 1
    return n


def run_reference(variant: str, labels, x, tg, pct, tsz, n_steps: int = 3, lr: float = 1e-2, wd: float = 1e-3, act: str = "relu"):
    sys.path.insert(0, REF)
    for m in list(sys.modules):
        if m == "models" or m.startswith("models.") or m.startswith("activate_fun") or m.startswith("scheduler"):
            del sys.modules[m]
    mod = importlib.import_module(REF_MODULE[variant])
    novo = importlib.import_module("scheduler.novograd")
    n_class = len(labels) + 1
    model = mod.MyModel2(labels, 0.0, True)
    keys = list(model.state_dict().keys())
    shapes = R.state_shapes(variant, n_class)
    assert keys == [k for k, _ in shapes], "state_dict key order differs from oracle.state_shapes"
    for (k, shp), v in zip(shapes, model.state_dict().values()):
        assert tuple(v.shape) == tuple(shp), (k, v.shape, shp)
    model.load_state_dict(R.formula_state(variant, n_class))
    if act == "swish":
        n_swapped = swap_activation(model, importlib.import_module("activate_fun.Swish").Swish)
        assert n_swapped == len(R.block_table(variant)) * 2 + 2, n_swapped     # 2 per residual block + first_cnn + last_cnn2
    out = {}
    # eval-mode forward first (does not touch buffers)
    model.eval()
    with torch.no_grad():
        out["eval_logprobs"] = model(x, pct).numpy().copy()
    model.train()
    taps = {}
    hooks = []
    enc = model.encoder
    for name, sub in enc.named_children():
        def mk(nm):
            def hook(_m, _i, o):
                o0 = o[0] if isinstance(o, tuple) else o
                taps[nm] = o0.detach().clone()
            return hook
        hooks.append(sub.register_forward_hook(mk(name)))
    opt = novo.Novograd(model.parameters(), lr=lr, weight_decay=wd, betas=(0.8, 0.5))
    lossf = torch.nn.CTCLoss(blank=len(labels), reduction="none")
    losses = []
    for step in range(n_steps):
        opt.zero_grad()
        lp = model(x, pct)
        t_len = torch.mul(lp.size(1), pct).int()
        nll = lossf(lp.transpose(0, 1), tg, t_len, tsz)
        loss = torch.mean(nll)
        loss.backward()
        if step == 0:
            for h in hooks:
                h.remove()
            out["logprobs"] = lp.detach().numpy().copy()
            out["t_lengths"] = t_len.numpy().copy()
            out["nll"] = nll.detach().numpy().copy()
            out["argmax"] = lp.argmax(-1).to(torch.int16).numpy().copy()
            for nm, v in taps.items():
                out["tap_" + nm] = checksum(v)
            out["grad_norms"] = np.array([p.grad.norm().item() for p in model.parameters()])
            out["grad_sample"] = np.stack([checksum(p.grad) for p in model.parameters()])
            out["_grads"] = [p.grad.detach().clone() for p in model.parameters()]
            out["_lp"] = lp.detach().clone()
        losses.append(loss.item())
        opt.step()
        if step in (0, n_steps - 1):
            out["params_after_%d" % (step + 1)] = np.stack([checksum(p) for p in model.parameters()])
    out["losses"] = np.array(losses)
    sd = model.state_dict()
    out["running_after"] = np.stack([checksum(sd[k].float()) for k in sd if k.endswith(("running_mean", "running_var"))])
    out["_state_after"] = {k: v.detach().clone() for k, v in sd.items()}
    return out


def run_oracle(variant: str, labels, x, tg, pct, tsz, n_steps: int = 3, lr: float = 1e-2, wd: float = 1e-3, act: str = "relu"):
    n_class = len(labels) + 1
    m = R.OracleModel(variant, n_class, mask=True, act=act, state=R.formula_state(variant, n_class))
    out = {}
    m.training = False
    with torch.no_grad():
        out["eval_logprobs"] = m(x, pct).numpy().copy()
    st = R.NovogradState(len(m.parameters()))
    losses = []
    for step in range(n_steps):
        m.keep_taps = step == 0
        m.training = True
        m.requires_grad_(True)
        if step == 0:
            for p in m.parameters():
                p.grad = None
            lp = m(x, pct)
            t_len = R.mask_lengths(lp.size(1), pct)
            nll = R.ctc_loss_per_sample(lp, tg, t_len, tsz, n_class - 1)
            out["logprobs"] = lp.detach().numpy().copy()
            out["nll"] = nll.detach().numpy().copy()
            out["t_lengths"] = t_len.numpy().copy()
            for nm, v in m.taps.items():
                out["tap_" + nm] = checksum(v)
            # undo the buffer side effects of this extra forward so train_step starts clean
            m.state = R.formula_state(variant, n_class)
            m.taps = {}
            m.keep_taps = False
        loss, grads = R.train_step(m, st, x, tg, pct, tsz, lr, wd)
        if step == 0:
            out["_grads"] = grads
        losses.append(loss)
    out["losses"] = np.array(losses)
    out["_state_after"] = {k: v.detach().clone() for k, v in m.state.items()}
    return out


def check(a, b, what, rtol=2e-4, atol=2e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() if a.size else 0.0
    ref = np.abs(a).max() if a.size else 0.0
    ok = np.allclose(a, b, rtol=rtol, atol=atol)
    print("  %-34s max|d|=%.3e (max|ref|=%.3e) %s" % (what, err, ref, "ok" if ok else "MISMATCH"))
    assert ok, what


def lr_schedule_reference(n: int = 2600):
    sys.path.insert(0, REF)
    cos = importlib.import_module("scheduler.cosine_annearing_with_warmup")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sch = cos.CosineAnnealingWarmupRestarts(opt, first_cycle_steps=1200, cycle_mult=2, max_lr=1e-2,
                                                min_lr=1e-4, warmup_steps=1000, gamma=0.5)
        vals = []
        for _ in range(n):
            vals.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
    return np.array(vals)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    labels = [c.strip() for c in open(os.path.join(REF, "data", "labels.txt"))]
    assert len(labels) == 27
    x, tg, pct, tsz = golden_inputs()
    for variant in R.VARIANTS:
        print("variant", variant)
        ref = run_reference(variant, labels, x, tg, pct, tsz)
        ora = run_oracle(variant, labels, x, tg, pct, tsz)
        for k in ("eval_logprobs", "logprobs", "nll", "t_lengths", "losses"):
            check(ref[k], ora[k], k)
        for k in ref:
            if k.startswith("tap_"):
                nm = k[4:]
                nm_o = {"context_rnn": "context"}.get(nm, nm)
                check(ref[k], ora["tap_" + nm_o], k)
        # f32 round-off differences (e.g. our explicit LSTM loop vs ATen's) flip a few ReLU
        # gates and are amplified by the BN chain in backward, so gradients are compared in
        # relative L2 per tensor, not element-wise.
        worst = 0.0
        for i, (g_r, g_o) in enumerate(zip(ref["_grads"], ora["_grads"])):
            rel = ((g_r - g_o).norm() / (g_r.norm() + 1e-20)).item()
            worst = max(worst, rel)
            assert rel <= 5e-3, ("grad", i, rel)
        print("  grads ok (%d tensors, worst rel-L2 %.2e)" % (len(ref["_grads"]), worst))
        worst = 0.0
        for k, v in ref["_state_after"].items():
            o = ora["_state_after"][k].double()
            rel = ((v.double() - o).norm() / (v.double().norm() + 1e-20)).item()
            worst = max(worst, rel)
            assert rel <= 2e-2, ("state_after", k, rel)
        print("  state after 3 NovoGrad steps ok (worst rel-L2 %.2e)" % worst)
        save = {k: v for k, v in ref.items() if not k.startswith("_")}
        np.savez_compressed(os.path.join(GOLD, "model_%s.npz" % variant), **save)

    # Swish epilogues (north_star "BatchNorm + Swish"; activate_fun/Swish.py:9-10): the reference's own Swish module swapped into
    # the reference model's epilogues - pins the oracle's act="swish" path (forward, gradients, one NovoGrad step)
    for variant in ("plain", "context_se"):
        print("variant", variant, "/ swish epilogues")
        ref = run_reference(variant, labels, x, tg, pct, tsz, n_steps=1, act="swish")
        ora = run_oracle(variant, labels, x, tg, pct, tsz, n_steps=1, act="swish")
        for k in ("eval_logprobs", "logprobs", "nll", "t_lengths", "losses"):
            check(ref[k], ora[k], k)
        worst = 0.0
        for i, (g_r, g_o) in enumerate(zip(ref["_grads"], ora["_grads"])):
            rel = ((g_r - g_o).norm() / (g_r.norm() + 1e-20)).item()
            worst = max(worst, rel)
            assert rel <= 5e-3, ("grad", i, rel)
        print("  grads ok (%d tensors, worst rel-L2 %.2e)" % (len(ref["_grads"]), worst))
        save = {k: v for k, v in ref.items() if not k.startswith("_")}
        np.savez_compressed(os.path.join(GOLD, "model_%s_swish.npz" % variant), **save)

    # large-vocab short case (cfg5 shape class): C = 4334, plain model, 1 step
    vocab = [c.strip() for c in open(os.path.join(REF, "data", "aishell1-vocab.txt"))]
    x2, tg2, pct2, tsz2 = golden_inputs(B=2, T_in=81, S=6, V=len(vocab))
    pct2 = torch.tensor([1.0, 0.75])
    print("variant plain / aishell vocab", len(vocab))
    ref = run_reference("plain", vocab, x2, tg2, pct2, tsz2, n_steps=1)
    ora = run_oracle("plain", vocab, x2, tg2, pct2, tsz2, n_steps=1)
    for k in ("logprobs", "nll", "losses"):
        check(ref[k], ora[k], k)
    save = {k: v for k, v in ref.items() if not k.startswith("_") and k not in ("eval_logprobs", "logprobs")}
    save["logprob_checksum"] = checksum(ref["_lp"])
    np.savez_compressed(os.path.join(GOLD, "model_plain_aishell.npz"), **save)

    # LR schedule
    lr_ref = lr_schedule_reference()
    sch = R.CosineWarmupRestarts(1200, 2, 1e-2, 1e-4, 1000, 0.5)
    lr_o = []
    for _ in range(len(lr_ref)):
        lr_o.append(sch.lr)
        sch.step()
    check(lr_ref, np.array(lr_o), "lr schedule", rtol=1e-12, atol=1e-15)
    np.savez_compressed(os.path.join(GOLD, "lr_schedule.npz"), lr=lr_ref,
                        args=np.array([1200, 2, 1e-2, 1e-4, 1000, 0.5]))

    # mask-length truncation cases straight from torch semantics the reference uses
    cases = [(801, 0.3333333), (501, 0.7), (501, 1.0), (101, 0.9), (836, 0.123456)]
    lens = [int(torch.mul(T, torch.tensor([p], dtype=torch.float32)).int()) for T, p in cases]
    np.savez_compressed(os.path.join(GOLD, "mask_lengths.npz"), T=np.array([c[0] for c in cases]),
                        pct=np.array([c[1] for c in cases], dtype=np.float32), lens=np.array(lens))
    print("mask lengths", lens)
    print("golden written to", GOLD)


if __name__ == "__main__":
    main()
