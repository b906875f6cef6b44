"""bf16-storage emulation of the CPU oracle + per-unit ("teacher-forced") runners.
TEST INFRASTRUCTURE ONLY (same import rule as ref_cpu.py: tests/, smoke(), bench.py's cpu_baseline leg).

The HIP path's bf16 mode (the bench dtype) keeps f32 master weights and f32 accumulators but STORES
activations, activation gradients and the GEMM / depthwise operands in bf16.  ``ref_cpu.OracleModel``
(pinned against the reference, f32) cannot tell a wrong bf16 kernel from bf16 rounding noise; the
subclass below rounds to bf16 at exactly the points where the plan in ``csrc/model.hip`` stores bf16
and computes everything between two stores in f64, so what is left between it and the GPU is the
accumulation order (f32 on the matrix cores) plus the rare bf16 rounding flips that causes.

Storage points (lightning_asr_amd/csrc/model.hip, bf16 mode):
  forward   feats | u = dw(x) | y = mask(pw(u)), y2 = res(x) | out = act(BN(y)[*se] + BN(y2)) | BiLSTM output
            weights: bf16 shadow for every 1x1 / LSTM-input / decoder GEMM and for the stride-1 depthwise taps
            (MFMA Toeplitz form); first_cnn's stride-2 depthwise taps stay f32; logits, log-probs, CTC are f32
  backward  d(logits) (bf16 GEMM operand; the bias gradient sums the f32 one) | d(out) | dy, dy2 | du | dx_res
            parameter gradients are f32 accumulations of bf16 operands and are never rounded
BN statistics are taken from the values as stored (the GEMM epilogue sums the rounded outputs).

Why per-unit runners: measured here, the end-to-end map of this BN stack is chaotic at bf16 resolution -
two emulations that differ only in f32-vs-f64 arithmetic BETWEEN the stores agree to 1.6e-4 on the loss
but only to 0.4 relative L2 on the gradients (CTC at initialisation sends an almost frame-constant
gradient into BatchNorm layers whose backward removes exactly that component).  A whole-step gradient
comparison can therefore not be tight whatever the kernels do.  ``run_unit`` / ``run_context`` /
``run_head`` take the GPU's OWN stored inputs of one unit (x_in, d(out)) and return what that unit must
produce from them (u, y, y2, out, dx, parameter gradients): same kernels, real sizes, no cross-layer
amplification, so the tolerance is a few bf16 flips.  With ``emulate=False`` the same code is the plain
f64 oracle of the unit (for the f32 parity mode).

The reference itself runs fp16 AMP (conf/conf.yaml:27-29): there is no bit-defined low-precision
reference, so this file is a *model of our own storage format* on top of the pinned f32 arithmetic.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ref_cpu as R


def rb(x: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even to bf16, kept in x's dtype"""
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


class _Store(torch.autograd.Function):
    """a bf16 tensor of the plan: value rounded forward, its gradient (another bf16 tensor) rounded backward"""
    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return rb(g)


class _Shadow(torch.autograd.Function):
    """bf16 shadow of an f32 master weight: rounded forward, gradient kept in full precision"""
    @staticmethod
    def forward(ctx, w):
        return rb(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _GradStore(torch.autograd.Function):
    """identity forward; the gradient flowing back through this edge is a bf16 tensor of the plan"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return rb(g)


class _RoundFwd(torch.autograd.Function):
    """value stored in bf16, gradient passed through unchanged (the roundings of ITS consumers are modelled where they happen)"""
    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _ActForcedMask(torch.autograd.Function):
    """ReLU whose derivative mask is given (the GPU's own `out > 0`): an element with |z| ~ 1e-7 may take either sign
    under two f32 evaluation orders, and ONE flipped element of 8.2 M is 3.5e-4 of the gradient's L2 norm and ~1e-2 of
    its channel's beta-gradient (measured: units with no such element agree to 1e-4 / 2e-8, units with 1-4 to 1e-3)."""
    @staticmethod
    def forward(ctx, z, mask):
        ctx.save_for_backward(mask)
        return torch.relu(z)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


class Bf16OracleModel(R.OracleModel):
    """OracleModel with the bf16 storage points of the HIP plan.  ``dtype`` is the arithmetic between two
    stores (float64 by default: the statistics the kernels reduce in f64 block partials stay exact).
    ``emulate=False`` switches every rounding off (plain oracle in ``dtype``)."""

    def __init__(self, variant: str, n_class: int, mask: bool = True, act: str = "relu",
                 state: Optional[Dict[str, torch.Tensor]] = None, in_c: int = 64, dtype=torch.float64, emulate: bool = True):
        super().__init__(variant, n_class, mask, act, state, in_c)
        self.dtype, self.emulate = dtype, emulate
        self.drop = None              # (keep mask (B, C, T) of the unit being run, 1/(1-p)): the GPU's own mask (run_unit)
        self.lean_head = False        # large-vocabulary head of csrc/ctc_lean.hip: the logits themselves are a bf16 tensor
        self.forced_mask: Optional[torch.Tensor] = None      # consumed by the next activation (run_unit)
        self.forced_se_mask: Optional[torch.Tensor] = None   # (B, C/8) derivative mask of the SE MLP's inner ReLU, consumed by the next _sep
        for k, v in self.state.items():
            if v.is_floating_point():
                # BatchNorm always in f64: its backward subtracts the two largest components of the incoming gradient, and
                # the kernels reduce those sums in f64 block partials
                self.state[k] = v.detach().to(torch.float64 if self._is_bn_key(k) else dtype)

    @staticmethod
    def _is_bn_key(k: str) -> bool:
        return (".bn." in k or ".reside.1." in k or ".last_cnn2.1." in k) and not k.endswith("num_batches_tracked")

    def _bn(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        s = self.state
        if self.training:
            s[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x.double(), s[prefix + ".running_mean"], s[prefix + ".running_var"], s[prefix + ".weight"],
                            s[prefix + ".bias"], self.training, R.BN_MOM, R.BN_EPS).to(x.dtype)

    def _act(self, z: torch.Tensor) -> torch.Tensor:
        if self.forced_mask is not None and self.act == "relu":
            m, self.forced_mask = self.forced_mask, None
            return _ActForcedMask.apply(z, m)
        return R.activation(z, self.act)

    def _dropout(self, z):
        """nn.Dropout at the end of SeprationConv / last_cnn2 (models/QuartNet.py:38,149) with a GIVEN mask"""
        if self.drop is None:
            return z
        keep, inv = self.drop
        return z * (keep.to(z.dtype) * inv)

    def st(self, x):
        return _Store.apply(x) if self.emulate else x

    def sh(self, w):
        return _Shadow.apply(w) if self.emulate else w

    def gs(self, x):
        return _GradStore.apply(x) if self.emulate else x

    def _sep(self, x, lens, prefix: str, k: int, stride: int, last: bool) -> torch.Tensor:
        s = self.state
        ci = x.size(1)
        w_dw = s[prefix + ".depthwise_conv.weight"]
        if stride == 1:                              # MFMA Toeplitz form: packed bf16 taps (csrc/conv.hip)
            w_dw = self.sh(w_dw)
        u = self.st(F.conv1d(x, w_dw, None, stride, k // 2, 1, ci))
        y = F.conv1d(u, self.sh(s[prefix + ".pointwise_conv.weight"]))
        if self.mask:
            y = R._time_mask(y, lens)
        y = self.st(y)
        if self.keep_taps:
            self.taps[prefix + ".u"], self.taps[prefix + ".y"] = u, y
        z = self._bn(y, prefix + ".bn")
        if self.variant == "context_se":
            pooled = z.mean(dim=2)
            h = F.linear(pooled, s[prefix + ".se.fc.0.weight"])
            if self.forced_se_mask is not None:      # the SE MLP's own ReLU (models/QuartNetContextSE.py:14) has the same near-zero
                sm, self.forced_se_mask = self.forced_se_mask, None     # ambiguity as the unit's: B x C/8 values, ONE flip = 3e-2 of dW1
                h = _ActForcedMask.apply(h, sm)
            else:
                h = F.relu(h)
            g = torch.sigmoid(F.linear(h, s[prefix + ".se.fc.2.weight"]))
            z = z * g.unsqueeze(2)
        if not last:
            return self.st(self._dropout(self._act(z)))
        return self._dropout(z)          # residual units: the main branch is dropped BEFORE the add (it ends the SeprationConv)

    def _block(self, x, lens, name: str, k: int) -> torch.Tensor:
        p = "encoder." + name
        main = self._sep(x, lens, p + ".seq.0", k, 1, True)
        y2 = self.st(F.conv1d(self.gs(x), self.sh(self.state[p + ".reside.0.weight"])))       # dx_res is its own bf16 tensor
        if self.keep_taps:
            self.taps[p + ".y2"] = y2
        res = self._bn(y2, p + ".reside.1")
        return self.st(self._act(main + res))

    def _last(self, x) -> torch.Tensor:
        y = self.st(F.conv1d(x, self.sh(self.state["encoder.last_cnn2.0.weight"])))
        if self.keep_taps:
            self.taps["encoder.last_cnn2.y"] = y
        return self.st(self._dropout(self._act(self._bn(y, "encoder.last_cnn2.1"))))

    def _lstm_dir(self, x_btc: torch.Tensor, sfx: str, valid: torch.Tensor) -> torch.Tensor:
        """one direction of the BiLSTM (models/QuartNetContext.py:186-199) -> (B, H, T).  The gate pre-activations' input
        part is a GEMM on the bf16 shadow of W_ih with f32 output; its gradient dG is f32 for the recurrence and the bias
        sums and is cast to bf16 as the operand of the dW_ih / dx GEMMs; the recurrence runs on the f32 master weights."""
        s = self.state
        r = "encoder.context_rnn.rnn."
        B, T, _ = x_btc.shape
        H = R.LSTM_HIDDEN
        w_hh = s[r + "weight_hh_l0" + sfx]
        gx = self.gs(F.linear(x_btc, self.sh(s[r + "weight_ih_l0" + sfx]))) + (s[r + "bias_ih_l0" + sfx] + s[r + "bias_hh_l0" + sfx])
        h = gx.new_zeros(B, H)
        c = gx.new_zeros(B, H)
        out = [None] * T
        for t in (range(T) if sfx == "" else range(T - 1, -1, -1)):
            g = gx[:, t] + F.linear(h, w_hh)
            i, f, gg, o = g.split(H, dim=1)
            c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h_new = torch.sigmoid(o) * torch.tanh(c_new)
            m = valid[:, t:t + 1]
            c = m * c_new + (1 - m) * c
            h = m * h_new + (1 - m) * h
            out[t] = m * h_new
        return torch.stack(out, dim=2)

    def _bilstm(self, x: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        """x (B, T, 256) -> (B, T, 80), stored in bf16 (written straight into the 336-channel cat buffer)"""
        valid = (torch.arange(x.size(1)).unsqueeze(0) < lens.unsqueeze(1)).to(x.dtype)
        out = torch.cat([self._lstm_dir(x, sfx, valid) for sfx in ("", "_reverse")], dim=1)
        return self.st(out).transpose(1, 2)

    def encode(self, inputs: torch.Tensor, pct: torch.Tensor) -> torch.Tensor:
        x = inputs.to(self.dtype).squeeze(1)
        if self.emulate:
            x = rb(x)                                # the mel kernel writes bf16 features
        T1 = (x.size(2) + 2 * 16 - 33) // 2 + 1
        lens = R.mask_lengths(T1, pct)
        tap = self.taps.__setitem__ if self.keep_taps else (lambda k, v: None)
        x = self._sep(x, lens, "encoder.first_cnn", 33, 2, False)
        tap("first_cnn", x)
        for name, ci, co, k in R.block_table(self.variant):
            if name == "block3" and self.variant != "plain":
                ctx = self._bilstm(x.transpose(1, 2), lens)
                tap("context", ctx)
                x = torch.cat((x, ctx.transpose(1, 2)), dim=1)
            x = self._block(x, lens, name, k)
            tap(name, x)
        x = self._last(x)
        tap("last_cnn2", x)
        return x

    def head(self, x: torch.Tensor) -> torch.Tensor:
        """decoder 1x1 + log_softmax: f32 logits from bf16 operands; backward casts d(logits) to bf16 for both
        decoder GEMMs while the bias gradient sums the f32 one"""
        logits = self.gs(F.conv1d(x, self.sh(self.state["decoder.weight"]))) + self.state["decoder.bias"].view(1, -1, 1)
        if self.lean_head and self.emulate:
            logits = _RoundFwd.apply(logits)      # stored bf16 (bias added in the GEMM epilogue before the rounding); softmax of the stored row
        if self.keep_taps:
            self.taps["logits"] = logits
        return F.log_softmax(logits.transpose(1, 2), dim=-1)

    def forward(self, inputs: torch.Tensor, pct: torch.Tensor) -> torch.Tensor:
        return self.head(self.encode(inputs, pct))

    __call__ = forward


def loss_and_grads(model: R.OracleModel, inputs, targets, pct, target_sizes):
    """fwd + mean CTC + bwd (no optimiser step).  Returns (loss, nll (B), log-probs, [grads in parameter order])."""
    model.training = True
    model.requires_grad_(True)
    params = model.parameters()
    for p in params:
        p.grad = None
    lp = model.forward(inputs, pct)
    tl = R.mask_lengths(lp.size(1), pct)
    nll = R.ctc_loss_per_sample(lp, targets, tl, target_sizes, blank=model.n_class - 1)
    loss = nll.mean()
    loss.backward()
    grads = [p.grad.detach().clone() for p in params]
    model.requires_grad_(False)
    return float(loss.detach()), nll.detach(), lp.detach(), grads


# ------------------------------------------------------------------------------------------------------------
# Per-unit runners.  Tensors are (B, C, T) like the reference modules; dtype = model.dtype.
# ------------------------------------------------------------------------------------------------------------
def _prefixes(variant: str, unit: str) -> List[str]:
    """state_dict key prefixes owned by a unit of the plan ('first_cnn', 'block*', 'last_cnn2')"""
    if unit == "first_cnn":
        return ["encoder.first_cnn."]
    if unit == "last_cnn2":
        return ["encoder.last_cnn2."]
    return ["encoder." + unit + "."]


def _unit_params(model: R.OracleModel, prefixes) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in model.state.items() if not R.is_buffer(k) and any(k.startswith(p) for p in prefixes)}


def run_unit(model: Bf16OracleModel, unit: str, x_in: torch.Tensor, lens: torch.Tensor, dout: Optional[torch.Tensor],
             act_mask: Optional[torch.Tensor] = None, drop=None, se_mask: Optional[torch.Tensor] = None):
    """One unit of the plan from ITS OWN stored input: forward (training-mode BN) and, if ``dout`` is given, backward.
    act_mask: the GPU's `out > 0` (ReLU derivative; see _ActForcedMask).
    Returns {"u","y","y2","out","dx", "grads": {key: tensor}} (absent entries omitted)."""
    model.training, model.keep_taps, model.taps = True, True, {}
    model.forced_mask = act_mask
    model.forced_se_mask = se_mask    # the GPU's own `se_hidden > 0` (B, C/8), SE units only
    model.drop = drop                 # (keep mask (B, C, T), 1/(1-p)) or None
    params = _unit_params(model, _prefixes(model.variant, unit))
    for p in params.values():
        p.requires_grad_(True)
        p.grad = None
    x = x_in.to(model.dtype).detach().clone().requires_grad_(dout is not None and unit != "first_cnn")
    if unit == "first_cnn":
        out = model._sep(x, lens, "encoder.first_cnn", 33, 2, False)
        pfx = "encoder.first_cnn"
    elif unit == "last_cnn2":
        out = model._last(x)
        pfx = "encoder.last_cnn2"
    else:
        k = {n: k_ for n, _, _, k_ in R.block_table(model.variant)}[unit]
        out = model._block(x, lens, unit, k)
        pfx = "encoder." + unit + ".seq.0"
    res = {"out": out.detach()}
    for name, key in (("u", pfx + ".u"), ("y", pfx + ".y"), ("y2", "encoder." + unit + ".y2")):
        if key in model.taps:
            res[name] = model.taps[key].detach()
    if dout is not None:
        out.backward(dout.to(model.dtype))
        if x.grad is not None:
            res["dx"] = rb(x.grad) if model.emulate else x.grad
        res["grads"] = {k: p.grad.detach().clone() for k, p in params.items()}
    for p in params.values():
        p.requires_grad_(False)
        p.grad = None
    model.keep_taps, model.taps, model.drop = False, {}, None
    model.forced_mask = model.forced_se_mask = None
    return res


def run_context(model: Bf16OracleModel, x23: torch.Tensor, lens: torch.Tensor, dcat: Optional[torch.Tensor]):
    """BiLSTM context branch + channel cat (models/QuartNetContext.py:171-199) from the stored block23 output (B,256,T).
    Backward as the plan composes it: d(x23) = store(store(d(cat)[:256] + dG_f W_ih_f) + dG_r W_ih_r)."""
    s = model.state
    r = "encoder.context_rnn.rnn."
    params = {k: v for k, v in s.items() if k.startswith(r)}
    for p in params.values():
        p.requires_grad_(True)
        p.grad = None
    xs = [x23.to(model.dtype).detach().clone().requires_grad_(dcat is not None) for _ in range(2)]
    valid = (torch.arange(x23.shape[2]).unsqueeze(0) < lens.unsqueeze(1)).to(model.dtype)
    outs = [model._lstm_dir(xs[d].transpose(1, 2), sfx, valid) for d, sfx in enumerate(("", "_reverse"))]
    ctx = model.st(torch.cat(outs, dim=1))            # (B, 80, T)
    res = {"ctx": ctx.detach()}
    if dcat is not None:
        dcat = dcat.to(model.dtype)
        ctx.backward(dcat[:, 256:])
        g = dcat[:, :256] + xs[0].grad
        if model.emulate:
            g = rb(g)
        g = g + xs[1].grad
        res["dx"] = rb(g) if model.emulate else g
        res["grads"] = {k: p.grad.detach().clone() for k, p in params.items()}
    for p in params.values():
        p.requires_grad_(False)
        p.grad = None
    return res


def run_head(model: Bf16OracleModel, x_last: torch.Tensor, pct: torch.Tensor, targets, target_sizes,
             glogits_in: Optional[torch.Tensor] = None):
    """decoder + log_softmax + mean CTC (+ backward) from the stored last_cnn2 output (B,1024,T).
    glogits_in (B,T,C): the GPU's own d(loss)/d(logits); the decoder backward (dx, parameter gradients) then starts from it
    (teacher-forced like the units), while "glogits" is still the oracle's own CTC gradient for comparison.
    Returns {"logp","nll","loss","glogits" (B,T,C), "dx", "grads"}."""
    keys = ["decoder.weight", "decoder.bias"]
    for k in keys:
        model.state[k].requires_grad_(True)
        model.state[k].grad = None
    x = x_last.to(model.dtype).detach().clone().requires_grad_(True)
    model.keep_taps, model.taps = True, {}
    lp = model.head(x)
    logits = model.taps["logits"]
    tl = R.mask_lengths(lp.size(1), pct)
    nll = R.ctc_loss_per_sample(lp, targets, tl, target_sizes, blank=model.n_class - 1)
    loss = nll.mean()
    (g_own,) = torch.autograd.grad(loss, logits, retain_graph=glogits_in is not None)
    logits.backward(g_own if glogits_in is None else glogits_in.to(model.dtype).transpose(1, 2))
    res = {"logp": lp.detach(), "nll": nll.detach(), "loss": float(loss.detach()), "glogits": g_own.transpose(1, 2).detach(),
           "dx": rb(x.grad) if model.emulate else x.grad, "grads": {k: model.state[k].grad.detach().clone() for k in keys}}
    for k in keys:
        model.state[k].requires_grad_(False)
        model.state[k].grad = None
    model.keep_taps, model.taps = False, {}
    return res
