"""Python handle on the native execution plan (lasr_model_* in include/lasr.h): flat f32 parameter /
buffer / gradient storage on the GPU plus per-tensor views named like the reference state_dict."""
from __future__ import annotations

import os

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import call
from .ops import _p, _stream, torch_dtype


class TensorInfo:
    __slots__ = ("name", "shape", "kind", "offset", "numel")

    def __init__(self, name, shape, kind, offset):
        self.name, self.shape, self.kind, self.offset = name, tuple(shape), kind, offset
        n = 1
        for s in shape:
            n *= s
        self.numel = n


class NativeModel:
    """Owns the lasr_model_t handle and the caller-side buffers the C ABI works on."""

    def __init__(self, variant: str, n_class: int, mask: bool = True, act: str = "relu", dtype=torch.float32,
                 in_c: int = 64, device="cuda"):
        lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LasrError("NativeModel runs on the GPU only (hand-written HIP); no CPU fallback exists")
        if self.device.index is None:              # "cuda" -> the current device, as an indexed device
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.cfg = _lib.ModelConfig(_lib.VARIANT[variant], n_class, in_c, int(mask), {"relu": 1, "swish": 2}[act],
                                    _lib.F32 if dtype == torch.float32 else _lib.BF16)
        self.variant, self.n_class, self.act_dtype = variant, n_class, dtype
        h = C.c_void_p()
        _lib.check(lib.lasr_model_create(C.byref(self.cfg), C.byref(h)), "lasr_model_create")
        self._h = h
        self._lib = lib
        self.tensors: List[TensorInfo] = []
        n = lib.lasr_model_tensor_info(h, -1, None, 0, None, None, None, None)
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        ndim, kind, off = C.c_int32(), C.c_int32(), C.c_int64()
        for i in range(n):
            lib.lasr_model_tensor_info(h, i, name, 256, shape, C.byref(ndim), C.byref(kind), C.byref(off))
            self.tensors.append(TensorInfo(name.value.decode(), [shape[d] for d in range(ndim.value)], kind.value, off.value))
        self.n_param = lib.lasr_model_param_elems(h)
        self.n_buffer = lib.lasr_model_buffer_elems(h)
        self.params = torch.zeros(self.n_param, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.n_param, dtype=torch.float32, device=self.device)
        self.buffers = torch.zeros(max(self.n_buffer, 1), dtype=torch.float32, device=self.device)
        self.counters: Dict[str, torch.Tensor] = OrderedDict()   # num_batches_tracked (host side, int64)
        self._bump = 0          # training forwards replayed from a captured graph (num_batches_tracked is applied lazily)
        self._ws: Optional[torch.Tensor] = None
        self._ws_key: Optional[Tuple[int, int, int]] = None
        self._last_feats = None
        self._last_logp = None
        for t in self.tensors:
            if t.kind == 1 and t.name.endswith("running_var"):
                self.buffers[t.offset:t.offset + t.numel] = 1.0
            if t.kind == 2:
                self.counters[t.name] = torch.zeros((), dtype=torch.int64)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.lasr_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- state --------------------------------------------------------------------------------
    def param_infos(self) -> List[TensorInfo]:
        return [t for t in self.tensors if t.kind == 0]

    def view(self, t: TensorInfo, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        base = flat if flat is not None else (self.params if t.kind == 0 else self.buffers)
        return base[t.offset:t.offset + t.numel].view(t.shape)

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        sd = OrderedDict()
        self._apply_bump()
        for t in self.tensors:
            sd[t.name] = self.counters[t.name].clone() if t.kind == 2 else self.view(t).detach().clone()
        return sd

    def set_dropout(self, p: float, seed: int = 0) -> None:
        """nn.Dropout(p) of every SeprationConv and of last_cnn2 (models/QuartNet.py:26,38,149) in training forwards: a
        counter-based mask regenerated in forward and backward (csrc/dropout.h).  p = 0 switches it off."""
        if not 0.0 <= p < 1.0:
            raise ValueError("dropout probability has to be in [0, 1), got %r" % (p,))
        if not hasattr(self, "drop_step"):
            self.drop_step = torch.zeros(1, dtype=torch.int64, device=self.device)     # index of the current training forward
        self.drop_p, self.drop_seed = float(p), int(seed) & 0xFFFFFFFFFFFFFFFF
        call("lasr_model_set_dropout", self._h, float(p), self.drop_seed, _p(self.drop_step))

    def dropout_mask(self, unit: int, n_elems: int) -> torch.Tensor:
        """keep-mask (uint8, 1 = kept) unit `unit` drew in the LAST training forward, for verification against a CPU reference"""
        d = _lib.Dropout(self.drop_step.data_ptr(), self.drop_seed, unit, self.drop_p)
        out = torch.empty(n_elems, dtype=torch.uint8, device=self.device)
        call("lasr_dropout_mask", C.byref(d), n_elems, _p(out), _stream())
        return out

    def bump_counters(self, n: int = 1) -> None:
        """n training forwards ran without passing through forward() / loss_backward() (graph replays)"""
        self._bump += n

    def _apply_bump(self) -> None:
        if self._bump:
            for k in self.counters:
                self.counters[k] += self._bump
            self._bump = 0

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        self._bump = 0
        names = {t.name for t in self.tensors}
        if strict:
            missing, extra = names - set(sd), set(sd) - names
            if missing or extra:
                raise KeyError("state_dict mismatch: missing %s unexpected %s" % (sorted(missing)[:5], sorted(extra)[:5]))
        with torch.no_grad():
            for t in self.tensors:
                if t.name not in sd:
                    continue
                v = sd[t.name]
                if t.kind == 2:
                    self.counters[t.name] = v.detach().to("cpu", torch.int64).clone()
                else:
                    if tuple(v.shape) != t.shape:
                        raise ValueError("%s: shape %s != %s" % (t.name, tuple(v.shape), t.shape))
                    self.view(t).copy_(v.to(self.device, torch.float32))

    def init_parameters(self, seed: int = 0) -> None:
        """PyTorch-default initialisation of the reference modules (Conv1d/Linear/LSTM:
        U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BatchNorm: weight 1, bias 0, mean 0, var 1)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for t in self.tensors:
                if t.kind == 2:
                    self.counters[t.name].zero_()
                elif t.kind == 1:
                    self.view(t).fill_(1.0 if t.name.endswith("running_var") else 0.0)
                elif ".bn." in t.name or ".reside.1." in t.name or ".last_cnn2.1." in t.name:
                    self.view(t).fill_(1.0 if t.name.endswith("weight") else 0.0)
                else:
                    if "rnn" in t.name:
                        fan_in = 40
                    elif t.name == "decoder.bias":
                        fan_in = 1024
                    else:
                        fan_in = 1
                        for s_ in t.shape[1:]:
                            fan_in *= s_
                    bound = 1.0 / (fan_in ** 0.5)
                    self.view(t).copy_(((torch.rand(t.shape, generator=g) * 2 - 1) * bound).to(self.device))

    def arm_prefetch(self, wave, sample_lens=None, dither=None, aug=None, normalize: bool = True, out=None, logical_len=None):
        """The next loss_backward / loss_backward_staged call also computes the log-mel features of `wave` (B, L) - the NEXT
        step's batch - in the grid of its CTC lattice kernel (lasr_ctc_loss_mel).  Returns (feats (B, T, 64), pct (B)), valid
        once that call has been enqueued."""
        B, P = wave.shape
        L = int(logical_len) if logical_len is not None else P      # (rows wider than the longest utterance: ops.mel)
        T = int(self._lib.lasr_mel_num_frames(L))
        if out is not None:                       # caller-owned (feats (B, T, 64), pct (B)): e.g. the ping-pong pair of two captured graphs
            feats, pct = out
            if tuple(feats.shape) != (B, T, 64) or feats.dtype != self.act_dtype or not feats.is_contiguous():
                raise ValueError("arm_prefetch(out=): feats must be a contiguous (%d, %d, 64) %s tensor" % (B, T, self.act_dtype))
        else:
            feats = torch.empty(B, T, 64, dtype=self.act_dtype, device=self.device)
            pct = torch.empty(B, dtype=torch.float32, device=self.device)
        frames = torch.empty(B, dtype=torch.int32, device=self.device)
        nb = int(self._lib.lasr_mel_workspace_bytes(B, T))
        ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=self.device)
        from .ops import wave_src
        src = wave_src(wave, dither, P if P != L else 0)
        call("lasr_model_set_prefetch_src", self._h, C.byref(src), _p(sample_lens), _p(aug), B, L, int(normalize), _p(feats),
             _lib.F32 if self.act_dtype == torch.float32 else _lib.BF16, _p(frames), _p(pct), _p(ws), nb)
        self._prefetch_keep = (wave, sample_lens, dither, aug, frames, ws)       # alive until the call consumed them
        return feats, pct

    def disarm_prefetch(self) -> None:
        """forget an armed feature prefetch (a capture that failed between arm_prefetch and its loss call)"""
        self._prefetch_keep = None
        call("lasr_model_clear_prefetch", self._h)

    def bucket_bounds(self) -> List[int]:
        """Element offsets cutting the flat gradient into the all-reduce buckets of SURVEY §8e, in layer
        order: first_cnn | 256-channel blocks | 512-channel blocks | last_cnn2 + (context) + decoder."""
        cuts = [0]
        group = None
        for t in self.param_infos():
            n = t.name
            if n.startswith("encoder.first_cnn"):
                g = 0
            elif n.startswith(("encoder.block1", "encoder.block2")):
                g = 1
            elif n.startswith(("encoder.block3", "encoder.block4", "encoder.block5", "encoder.block6")):
                g = 2
            else:
                g = 3
            if group is not None and g != group:
                cuts.append(t.offset)
            group = g
        cuts.append(self.n_param)
        return cuts

    def param_offsets(self) -> torch.Tensor:
        offs = [t.offset for t in self.param_infos()] + [self.n_param]
        return torch.tensor(offs, dtype=torch.int64, device=self.device)

    # ---- execution ------------------------------------------------------------------------------
    @property
    def lean_head(self) -> bool:
        """large-vocabulary bf16 head available (bf16 logits + row statistics instead of f32 log-probs; csrc/ctc_lean.hip)"""
        return self.act_dtype == torch.bfloat16 and 256 <= self.n_class <= 9216 and not os.environ.get("LASR_NO_LEAN_HEAD")

    def _logp_buffer(self, B: int, T: int, want_logp: bool):
        if not want_logp and self.lean_head:
            return None                      # the library then runs the lean head: no (B, T', C) f32 tensor is written or read
                                             # (the workspace still reserves the dense head's f32 logits for eval forwards / want_logp=True)
        return torch.empty(B, T, self.n_class, dtype=torch.float32, device=self.device)

    def out_frames(self, T_in: int) -> int:
        return int(self._lib.lasr_model_out_frames(self._h, T_in))

    def workspace(self, B: int, T_in: int, S_max: int = 1) -> torch.Tensor:
        key = (B, T_in, max(S_max, 1))
        if self._ws is None or self._ws_key != key:
            nb = self._lib.lasr_model_workspace_bytes(self._h, B, T_in, key[2])
            if self._ws is None or self._ws.numel() < nb:
                self._ws = None
                self._ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        return self._ws

    def tap(self, name: str) -> torch.Tensor:
        """Intermediate of the last forward, as a [B][T][C] view of the workspace."""
        B, T_in, S = self._ws_key
        shape = (C.c_int64 * 3)()
        off = self._lib.lasr_model_tap(self._h, name.encode(), B, T_in, S, shape)
        if off < 0:
            raise KeyError(name)
        shp = tuple(shape[i] for i in range(3))
        if name in ("logits", "grad_logits", "lse") or name.endswith(".se_hidden"):
            dt = torch.float32
        elif name == "lens":
            return self._ws[off:off + 4 * B].view(torch.int32)
        elif name.startswith("bwd."):
            # [N][c] gradient at the head of an [N][cmax] allocation: the caller slices the flat view
            n = shp[0] * shp[1] * shp[2] * (4 if self.act_dtype == torch.float32 else 2)
            return self._ws[off:off + n].view(self.act_dtype)
        else:
            dt = self.act_dtype
        n = shp[0] * shp[1] * shp[2] * (4 if dt == torch.float32 else 2)
        return self._ws[off:off + n].view(dt).view(shp)

    def forward(self, feats_btc: torch.Tensor, pct: torch.Tensor, training: bool = True, want_argmax: bool = True):
        """feats [B][T_in][in_c] (act dtype), pct (B) f32 -> (logp (B,T',C) f32, argmax (B,T') i32 | None)."""
        B, T_in, _ = feats_btc.shape
        if feats_btc.dtype != self.act_dtype:
            raise TypeError("features must be %s" % self.act_dtype)
        ws = self.workspace(B, T_in, self._ws_key[2] if self._ws_key and self._ws_key[:2] == (B, T_in) else 1)
        T = self.out_frames(T_in)
        logp = torch.empty(B, T, self.n_class, dtype=torch.float32, device=self.device)
        am = torch.empty(B, T, dtype=torch.int32, device=self.device) if want_argmax else None
        call("lasr_model_forward", self._h, _p(self.params), _p(self.buffers), _p(feats_btc), _p(pct), B, T_in, int(training),
             _p(logp), _p(am), _p(ws), ws.numel(), _stream())
        if training:
            for k in self.counters:
                self.counters[k] += 1
            self._last_feats, self._last_logp = feats_btc, logp
        return logp, am

    def backward(self, grad_logp: torch.Tensor) -> torch.Tensor:
        """dL/d(log-probs) -> flat gradient buffer (views via ``view(t, self.grads)``)."""
        if self._last_feats is None:
            raise RuntimeError("backward() needs a preceding training forward()")
        B, T_in, _ = self._last_feats.shape
        ws = self._ws
        call("lasr_model_backward", self._h, _p(self.params), _p(self._last_feats), _p(self._last_logp),
             _p(grad_logp.contiguous()), B, T_in, _p(self.grads), _p(ws), ws.numel(), _stream())
        return self.grads

    def unit_names(self) -> List[str]:
        n = self._lib.lasr_model_num_units(self._h)
        buf = C.create_string_buffer(128)
        out = []
        for i in range(n):
            _lib.check(self._lib.lasr_model_unit_info(self._h, i, buf, 128), "lasr_model_unit_info")
            out.append(buf.value.decode())
        return out

    def bucket_schedule(self, n_buckets: Optional[int] = None):
        """[(unit_stop, [(lo, hi), ...])] from the LAST bucket to the first: once the backward stage ending at ``unit_stop`` is
        enqueued, those pieces of the flat gradient are final and their all-reduce may start.

        n_buckets (default: env LASR_DP_BUCKETS, else 2): 4 = SURVEY 8e's cuts (last_cnn2+decoder | 512-channel blocks |
        256-channel blocks | first_cnn); 2 = the first two and the last two merged.  Every stage ends in its own
        weight-gradient launch + reduction, so fewer stages compute faster (4 stages cost +0.18 ms per step on one GPU,
        2 stages +0.05 ms) while the big first bucket (17.8 MB) still rides under the second half of backward.
        Context / ContextSE: the BiLSTM's parameters sit between last_cnn2 and the decoder in the flat buffer but are
        differentiated with block3 (models/QuartNetContext.py:171-199), so with 4 buckets they travel with the 512-channel
        blocks as a second piece of that bucket; with 2 buckets the first stage ends at block3 and covers them anyway."""
        if n_buckets is None:
            n_buckets = int(os.environ.get("LASR_DP_BUCKETS", "2"))
        names = self.unit_names()
        bounds = self.bucket_bounds()                      # [0, |first_cnn|, |256 blocks|, |512 blocks|, n]
        first512 = next(i for i, n in enumerate(names) if n == "block3")
        last = len(names) - 1                              # last_cnn2 (+ decoder head)
        if n_buckets <= 1:
            return [(0, [(0, self.n_param)])]
        if n_buckets < 4:
            return [(first512, [(bounds[2], bounds[4])]), (0, [(bounds[0], bounds[2])])]
        lstm = [t for t in self.param_infos() if "context_rnn" in t.name]
        if lstm:
            l0, l1 = lstm[0].offset, lstm[-1].offset + lstm[-1].numel
            head = [(bounds[3], l0), (l1, bounds[4])]
            mid = [(bounds[2], bounds[3]), (l0, l1)]
        else:
            head, mid = [(bounds[3], bounds[4])], [(bounds[2], bounds[3])]
        return [(last, head), (first512, mid), (1, [(bounds[1], bounds[2])]), (0, [(bounds[0], bounds[1])])]

    def loss_backward_staged(self, feats_btc, pct, targets, tgt_lens, on_bucket, want_argmax: bool = True, n_buckets: Optional[int] = None,
                             want_logp: bool = True):
        """loss_backward in stages; ``on_bucket(ranges)`` (ranges = [(lo, hi), ...] of the flat gradient) is called right after
        the stage that finalises them has been enqueued (the data-parallel host starts that bucket's all-reduce there)."""
        B, T_in, _ = feats_btc.shape
        S = targets.shape[1]
        ws = self.workspace(B, T_in, S)
        T = self.out_frames(T_in)
        logp = self._logp_buffer(B, T, want_logp)
        am = torch.empty(B, T, dtype=torch.int32, device=self.device) if want_argmax else None
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        nll = torch.empty(B, dtype=torch.float32, device=self.device)
        sched = self.bucket_schedule(n_buckets)
        stop0, ranges0 = sched[0]
        call("lasr_model_loss_backward_partial", self._h, _p(self.params), _p(self.buffers), _p(feats_btc), _p(pct), _p(targets),
             _p(tgt_lens), B, T_in, S, _p(logp), _p(loss), _p(nll), _p(am), _p(self.grads), _p(ws), ws.numel(), stop0, _stream())
        on_bucket(ranges0)
        for stop, ranges in sched[1:]:
            call("lasr_model_backward_continue", self._h, _p(self.params), _p(feats_btc), B, T_in, _p(self.grads), _p(ws), ws.numel(),
                 stop, _stream())
            on_bucket(ranges)
        for k in self.counters:
            self.counters[k] += 1
        self._last_feats, self._last_logp = feats_btc, logp
        return loss, nll, logp, am

    def loss_backward_units(self, feats_btc, pct, targets, tgt_lens, on_unit, want_logp: bool = True):
        """loss_backward one unit per stage, from the last unit to the first; ``on_unit(i, name)`` runs after the stage of unit i
        has been enqueued: its parameter gradients are final, tap("bwd.g_prev") is d(output of unit i) and tap("bwd.g_cur")
        d(input of unit i).  Test hook for the per-unit parity checks (same kernels, same order as loss_backward)."""
        B, T_in, _ = feats_btc.shape
        S = targets.shape[1]
        ws = self.workspace(B, T_in, S)
        T = self.out_frames(T_in)
        logp = self._logp_buffer(B, T, want_logp)
        am = torch.empty(B, T, dtype=torch.int32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        nll = torch.empty(B, dtype=torch.float32, device=self.device)
        names = self.unit_names()
        last = len(names) - 1
        call("lasr_model_loss_backward_partial", self._h, _p(self.params), _p(self.buffers), _p(feats_btc), _p(pct), _p(targets),
             _p(tgt_lens), B, T_in, S, _p(logp), _p(loss), _p(nll), _p(am), _p(self.grads), _p(ws), ws.numel(), last, _stream())
        on_unit(last, names[last])
        for i in range(last - 1, -1, -1):
            call("lasr_model_backward_continue", self._h, _p(self.params), _p(feats_btc), B, T_in, _p(self.grads), _p(ws), ws.numel(),
                 i, _stream())
            on_unit(i, names[i])
        for k in self.counters:
            self.counters[k] += 1
        self._last_feats, self._last_logp = feats_btc, logp
        return loss, nll, logp, am

    def loss_backward(self, feats_btc, pct, targets, tgt_lens, want_argmax: bool = True, want_logp: bool = True):
        """forward + mean CTC + backward.  Returns (loss (1), nll (B), logp, argmax).  want_logp=False: logp is None when the
        large-vocabulary head applies (bf16, C >= 256): no (B, T', C) f32 tensor is materialised."""
        B, T_in, _ = feats_btc.shape
        S = targets.shape[1]
        ws = self.workspace(B, T_in, S)
        T = self.out_frames(T_in)
        logp = self._logp_buffer(B, T, want_logp)
        am = torch.empty(B, T, dtype=torch.int32, device=self.device) if want_argmax else None
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        nll = torch.empty(B, dtype=torch.float32, device=self.device)
        call("lasr_model_loss_backward", self._h, _p(self.params), _p(self.buffers), _p(feats_btc), _p(pct), _p(targets),
             _p(tgt_lens), B, T_in, S, _p(logp), _p(loss), _p(nll), _p(am), _p(self.grads), _p(ws), ws.numel(), _stream())
        for k in self.counters:
            self.counters[k] += 1
        self._last_feats, self._last_logp = feats_btc, logp
        return loss, nll, logp, am
