"""Torch-tensor wrappers over the C ABI.  PyTorch is only the container for device memory and the
stream; every computation below is a liblasr.so (HIP, gfx950) call.  Tensors must live on the GPU:
there is no CPU path."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BF16, F32, call

ACT = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "swish": _lib.ACT_SWISH}


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("activations must be float32 or bfloat16, got %s" % t.dtype)


def torch_dtype(code: int):
    return torch.float32 if code == F32 else torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.LasrError("liblasr ops need GPU tensors (got a %s tensor); there is no CPU fallback" % t.device)
    if not t.is_contiguous():
        raise ValueError("liblasr ops need contiguous tensors")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------------- features
def mel_num_frames(n_samples: int) -> int:
    return int(_lib.load().lasr_mel_num_frames(n_samples))


class DeviceDither:
    """``y += 1e-5 * randn_like(y)`` (data_module.py:155) drawn inside the mel kernel: Philox keyed by ``seed``, counter =
    (sample / 4, utterance, step); ``step`` is a device scalar every mel call increments (a replayed graph draws fresh noise)."""

    def __init__(self, seed: int, device):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.step = torch.zeros(1, dtype=torch.int64, device=device)

    def noise(self, B: int, L: int) -> torch.Tensor:
        """the (B, L) N(0,1) noise the NEXT mel call draws (verification)"""
        out = torch.empty(B, L, dtype=torch.float32, device=self.step.device)
        call("lasr_dither_noise", self.seed, _p(self.step), B, L, _p(out), _stream())
        return out


def wave_src(wave: torch.Tensor, dither=None, pitch: int = 0) -> "_lib.WaveSrc":
    """lasr_wave_src for a (B, L) float32 or int16 (PCM) waveform tensor; dither: None | (B, L) f32 noise | DeviceDither;
    pitch: elements between rows when the rows are wider than the L the call is made with (0: L)"""
    if wave.dtype == torch.float32:
        wd = _lib.WAVE_F32
    elif wave.dtype == torch.int16:
        wd = _lib.WAVE_PCM16
    else:
        raise TypeError("waveforms must be float32 or int16 PCM, got %s" % wave.dtype)
    if isinstance(dither, DeviceDither):
        return _lib.WaveSrc(_p(wave), wd, None, dither.seed, _p(dither.step), int(pitch))
    if dither is not None and (dither.dtype != torch.float32 or tuple(dither.shape) != tuple(wave.shape)):
        raise TypeError("dither noise must be a float32 tensor of the waveform's shape")
    return _lib.WaveSrc(_p(wave), wd, _p(dither), 0, None, int(pitch))


def mel(wave: torch.Tensor, sample_lens: Optional[torch.Tensor] = None, dither=None,
        aug: Optional[torch.Tensor] = None, normalize: bool = True, dtype=torch.float32, want_bft: bool = True,
        want_btf: bool = True, out_btf: Optional[torch.Tensor] = None, out_pct: Optional[torch.Tensor] = None,
        logical_len: Optional[int] = None):
    """wave (B, L) f32 or int16 PCM -> (feats_bft (B,64,T) f32 | None, feats_btf (B,T,64) dtype | None, frames (B) i32, pct (B) f32).
    dither: None, a (B, L) N(0,1) tensor, or a DeviceDither (noise generated inside the kernel).
    logical_len: the batch's longest utterance when the rows of `wave` are wider than that (a loader's static row pitch): T = the
    frames of `logical_len` samples - the reference's pad-to-longest (data_module.py:222-248) - not of the row width."""
    B, P = wave.shape
    L = int(logical_len) if logical_len is not None else P
    if not 0 < L <= P:
        raise ValueError("logical_len %s outside the rows' width %d" % (logical_len, P))
    T = mel_num_frames(L)
    dev = wave.device
    bft = torch.empty(B, 64, T, dtype=torch.float32, device=dev) if want_bft else None
    btf = None
    if out_btf is not None:
        if tuple(out_btf.shape) != (B, T, 64) or out_btf.dtype != dtype or not out_btf.is_contiguous():
            raise ValueError("out_btf must be a contiguous (%d, %d, 64) %s tensor" % (B, T, dtype))
        btf = out_btf
    elif want_btf:
        btf = torch.empty(B, T, 64, dtype=dtype, device=dev)
    frames = torch.empty(B, dtype=torch.int32, device=dev)
    pct = out_pct if out_pct is not None else torch.empty(B, dtype=torch.float32, device=dev)
    nb = _lib.load().lasr_mel_workspace_bytes(B, T)
    ws = _ws(nb, dev)
    src = wave_src(wave, dither, P if P != L else 0)
    call("lasr_mel_fwd_src", _lib.C.byref(src), _p(sample_lens), _p(aug), B, L, int(normalize), _p(bft), _p(btf),
         F32 if dtype == torch.float32 else BF16, _p(frames), _p(pct), _p(ws), nb, _stream())
    return bft, btf, frames, pct


def bct_to_btc(x: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    B, Cc, T = x.shape
    out = torch.empty(B, T, Cc, dtype=dtype, device=x.device)
    call("lasr_bct_to_btc", _p(x), _p(out), _dt(out), B, Cc, T, _stream())
    return out


def btc_to_bct(x: torch.Tensor) -> torch.Tensor:
    B, T, Cc = x.shape
    out = torch.empty(B, Cc, T, dtype=torch.float32, device=x.device)
    call("lasr_btc_to_bct", _p(x), _dt(x), _p(out), B, Cc, T, _stream())
    return out


def mask_lengths(pct: torch.Tensor, T: int) -> torch.Tensor:
    lens = torch.empty(pct.numel(), dtype=torch.int32, device=pct.device)
    call("lasr_mask_lengths", _p(pct), pct.numel(), T, _p(lens), _stream())
    return lens


# ---------------------------------------------------------------------------------- conv pieces
def dwconv(x: torch.Tensor, w: torch.Tensor, stride: int = 1, flip: bool = False,
           addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B][T][C], w (C, k) or (C,1,k) f32."""
    B, Tin, Cc = x.shape
    k = w.shape[-1]
    Tout = (Tin + 2 * (k // 2) - k) // stride + 1
    y = torch.empty(B, Tout, Cc, dtype=x.dtype, device=x.device)
    call("lasr_dwconv_fwd", _p(x), _p(w), _p(addend), _p(y), _dt(x), B, Tin, Cc, k, stride, int(flip), _stream())
    return y


def dwconv_wgrad(x: torch.Tensor, dy: torch.Tensor, k: int, stride: int = 1) -> torch.Tensor:
    B, Tin, Cc = x.shape
    dw = torch.empty(Cc, k, dtype=torch.float32, device=x.device)
    nb = _lib.load().lasr_dwconv_wgrad_workspace_bytes(B, dy.shape[1], Cc, k)
    ws = _ws(nb, x.device)
    call("lasr_dwconv_wgrad", _p(x), _p(dy), _p(dw), _dt(x), B, Tin, Cc, k, stride, _p(ws), nb, _stream())
    return dw


def gemm(A: torch.Tensor, Bm: torch.Tensor, M: int, N: int, K: int, transA: bool = False, transB: bool = False,
         bias: Optional[torch.Tensor] = None, addend: Optional[torch.Tensor] = None,
         row_lens: Optional[torch.Tensor] = None, rows_per_seq: int = 0, want_stats: bool = False, split_k: int = 1,
         out_dtype=None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    out_dtype = out_dtype or A.dtype
    Cm = torch.empty(M, N, dtype=out_dtype, device=A.device)
    stats = torch.empty(2 * N, dtype=torch.float32, device=A.device) if want_stats else None
    nb = _lib.load().lasr_gemm_workspace_bytes(M, N, split_k, int(want_stats))
    ws = _ws(nb, A.device)
    call("lasr_gemm", _p(A), _p(Bm), _p(Cm), _dt(A), _dt(Cm), M, N, K, int(transA), int(transB), _p(bias), _p(addend),
         _p(row_lens), rows_per_seq, _p(stats), split_k, _p(ws), nb, _stream())
    return Cm, stats


def gemm_ld(A, lda, Bm, ldb, M, N, K, transA=False, transB=False, split_k: int = 1, out_dtype=None, ldc: int = 0):
    """lasr_gemm_ld: operands given as flat/padded buffers with explicit pitches; returns C (M, ldc or N)."""
    out_dtype = out_dtype or A.dtype
    Cm = torch.empty(M, ldc or N, dtype=out_dtype, device=A.device)
    nb = _lib.load().lasr_gemm_workspace_bytes(M, N, split_k, 0)
    ws = _ws(nb, A.device)
    call("lasr_gemm_ld", _p(A), lda, _p(Bm), ldb, _p(Cm), ldc, _dt(A), _dt(Cm), M, N, K, int(transA), int(transB), None, split_k,
         _p(ws), nb, _stream())
    return Cm


class _ReduceDesc(_lib.C.Structure):
    _fields_ = [("partials", _lib.C.c_void_p), ("out", _lib.C.c_void_p), ("n", _lib.C.c_int64), ("n_partials", _lib.C.c_int32)]


def reduce_many(segments):
    """segments: list of (partials (P_i, n_i) f32, out (n_i,) f32): out_i = partials_i.sum(0), one launch (<= 64)."""
    descs = (_ReduceDesc * len(segments))()
    for i, (pt, out) in enumerate(segments):
        descs[i].partials, descs[i].out, descs[i].n, descs[i].n_partials = _p(pt), _p(out), out.numel(), pt.shape[0]
    call("lasr_reduce_many", descs, len(segments), _stream())


def wgrad_multi(dys, xs, split_k: int = 4):
    """The 1x1 weight gradients dW_i = dy_i^T x_i of several layers in ONE split-K launch + ONE reduction
    (lasr_gemm_multi_split_partials + lasr_reduce_many).  dys[i] (rows, co_i), xs[i] (rows, ci_i) bf16."""
    n = len(dys)
    dev = dys[0].device
    probs = (_GemmProblem * n)()
    slabs = (_lib.C.c_void_p * n)()
    splits = (_lib.C.c_int * n)()
    outs, bufs = [], []
    for i in range(n):
        rows, co = dys[i].shape
        ci = xs[i].shape[1]
        out = torch.empty(co, ci, dtype=torch.float32, device=dev)
        buf = torch.empty(split_k, co * ci, dtype=torch.float32, device=dev)
        probs[i].A, probs[i].B, probs[i].C = _p(dys[i]), _p(xs[i]), _p(out)
        probs[i].M, probs[i].N, probs[i].K = co, ci, rows
        probs[i].bias = None; probs[i].row_lens = None; probs[i].rows_per_seq = 0; probs[i].stats = None
        slabs[i] = _p(buf)
        outs.append(out); bufs.append(buf)
    call("lasr_gemm_multi_split_partials", probs, n, split_k, slabs, splits, _stream())
    reduce_many([(bufs[i][:splits[i]], outs[i].view(-1)) for i in range(n)])
    return outs


def dwconv_bwd_fused(x: torch.Tensor, dy: torch.Tensor, w: torch.Tensor, addend: Optional[torch.Tensor] = None):
    """Depthwise backward of a stride-1 layer in one launch: (dW (C, k) f32, dx (B, T, C)) from x, dy (B, T, C) and the taps
    w (C, 1, k) / (C, k) f32 (lasr_dwconv_bwd_fused + lasr_reduce_many)."""
    B, T, Cc = x.shape
    k = w.shape[-1]
    dx = torch.empty_like(dy)
    nb = _lib.load().lasr_dwconv_wgrad_workspace_bytes(B, T, Cc, k)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    npart = _lib.C.c_int(0)
    call("lasr_dwconv_bwd_fused", _p(x), _p(dy), _p(w), _p(addend), _p(dx), _dt(x), B, T, Cc, k, _p(ws), nb, _lib.C.byref(npart), _stream())
    dw = torch.empty(Cc, k, dtype=torch.float32, device=x.device)
    reduce_many([(ws.view(torch.float32)[:npart.value * Cc * k].view(npart.value, Cc * k), dw.view(-1))])
    return dw, dx


class _FoldDesc(_lib.C.Structure):
    _fields_ = [("w", _lib.C.c_void_p), ("w_res", _lib.C.c_void_p), ("coef", _lib.C.c_void_p), ("coef2", _lib.C.c_void_p),
                ("w_out", _lib.C.c_void_p), ("bias_out", _lib.C.c_void_p), ("co", _lib.C.c_int64), ("ci", _lib.C.c_int64)]


def fold_bn_weights(w: torch.Tensor, coef: torch.Tensor, w_res: Optional[torch.Tensor] = None, coef2: Optional[torch.Tensor] = None):
    """Eval-mode BN folded into 1x1 conv weights: ([a W | a2 Wr] bf16 (co, ci (+ci)), b (+ b2) f32 (co)); coef = [a | b]."""
    co, ci = w.shape[0], w.shape[1]
    wout = torch.empty(co, ci * (2 if w_res is not None else 1), dtype=torch.bfloat16, device=w.device)
    bias = torch.empty(co, dtype=torch.float32, device=w.device)
    d = (_FoldDesc * 1)()
    d[0].w, d[0].w_res, d[0].coef, d[0].coef2 = _p(w), _p(w_res), _p(coef), _p(coef2)
    d[0].w_out, d[0].bias_out, d[0].co, d[0].ci = _p(wout), _p(bias), co, ci
    call("lasr_fold_bn_weights_many", d, 1, _stream())
    return wout, bias


def gemm_dual(a1: torch.Tensor, a2: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, row_lens: Optional[torch.Tensor] = None,
              rows_per_seq: int = 0, act: str = "relu") -> torch.Tensor:
    """act([a1 (rows past row_lens zeroed) | a2] @ w.T + bias): a1 (M, K1), a2 (M, K2), w (N, K1 + K2) bf16 -> (M, N) bf16."""
    M, K1 = a1.shape
    K2 = a2.shape[1]
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a1.device)
    call("lasr_gemm_dual", _p(a1), K1, _p(a2), K2, _p(w), _p(bias), _p(out), M, N, _p(row_lens), rows_per_seq, ACT[act], _stream())
    return out


def bn_finalize(stats, gamma, beta, running_mean, running_var, n_rows: int, eps: float = 1e-3, momentum: float = 0.1,
                training: bool = True):
    Cc = gamma.numel()
    coef = torch.empty(2 * Cc, dtype=torch.float32, device=gamma.device)
    saved = torch.empty(2 * Cc, dtype=torch.float32, device=gamma.device)
    call("lasr_bn_finalize", _p(stats), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(coef), _p(saved), Cc,
         n_rows, eps, momentum, int(training), _stream())
    return coef, saved


class _GemmProblem(_lib.C.Structure):
    _fields_ = [("A", _lib.C.c_void_p), ("B", _lib.C.c_void_p), ("C", _lib.C.c_void_p), ("M", _lib.C.c_int64), ("N", _lib.C.c_int64),
                ("K", _lib.C.c_int64), ("bias", _lib.C.c_void_p), ("row_lens", _lib.C.c_void_p), ("rows_per_seq", _lib.C.c_int64),
                ("stats", _lib.C.c_void_p)]


class _BnBranch(_lib.C.Structure):
    _fields_ = [("partials", _lib.C.c_void_p), ("n_partials", _lib.C.c_int), ("gamma", _lib.C.c_void_p), ("beta", _lib.C.c_void_p),
                ("running_mean", _lib.C.c_void_p), ("running_var", _lib.C.c_void_p), ("coef", _lib.C.c_void_p),
                ("saved", _lib.C.c_void_p), ("stats", _lib.C.c_void_p)]


def gemm_bn_fused(xs, ws_, bns, row_lens=None, rows_per_seq: int = 0, eps: float = 1e-3, momentum: float = 0.1):
    """A unit's 1x1 convolutions (1 or 2 problems: y_i = x_i @ w_i^T, the first optionally row-masked) with
    training-mode BN coefficients from the epilogue's partial sums in one reduce+finalize launch.
    bns[i] = (gamma, beta, running_mean, running_var).  Returns ([y_i], [coef_i], [saved_i], [stats_i])."""
    n = len(xs)
    dev = xs[0].device
    M = xs[0].shape[0]
    probs = (_GemmProblem * 2)()
    ys, coefs, saveds, stats = [], [], [], []
    for i in range(n):
        N, K = ws_[i].shape
        y = torch.empty(M, N, dtype=xs[i].dtype, device=dev)
        st = torch.empty(2 * N, dtype=torch.float32, device=dev)
        probs[i].A, probs[i].B, probs[i].C = _p(xs[i]), _p(ws_[i]), _p(y)
        probs[i].M, probs[i].N, probs[i].K = M, N, K
        probs[i].bias = None
        probs[i].row_lens = _p(row_lens) if (i == 0 and row_lens is not None) else None
        probs[i].rows_per_seq = rows_per_seq if i == 0 else 0
        probs[i].stats = _p(st)
        ys.append(y); stats.append(st)
    nb = sum(_lib.load().lasr_gemm_workspace_bytes(M, ws_[i].shape[0], 1, 1) for i in range(n))
    wsb = _ws(nb, dev)
    parts = (_lib.C.c_void_p * 2)()
    tiles = (_lib.C.c_int * 2)()
    call("lasr_gemm_batch_partials", probs, n, _dt(xs[0]), _dt(xs[0]), 0, 0, _p(wsb), wsb.numel(), parts, tiles, _stream())
    brs = (_BnBranch * 2)()
    for i in range(n):
        N = ws_[i].shape[0]
        g, b, rm, rv = bns[i]
        coef = torch.empty(2 * N, dtype=torch.float32, device=dev)
        saved = torch.empty(2 * N, dtype=torch.float32, device=dev)
        brs[i].partials, brs[i].n_partials = parts[i], tiles[i]
        brs[i].gamma, brs[i].beta, brs[i].running_mean, brs[i].running_var = _p(g), _p(b), _p(rm), _p(rv)
        brs[i].coef, brs[i].saved, brs[i].stats = _p(coef), _p(saved), _p(stats[i])
        coefs.append(coef); saveds.append(saved)
    call("lasr_bn_finalize_partials", brs, n, ws_[0].shape[0], M, eps, momentum, _stream())
    return ys, coefs, saveds, stats


def bn_act(y, coef, y2=None, coef2=None, se_scale=None, act: str = "relu") -> torch.Tensor:
    B, T, Cc = y.shape
    out = torch.empty_like(y)
    call("lasr_bn_act_fwd", _p(y), _p(coef), _p(y2), _p(coef2), _p(se_scale), _p(out), _dt(y), B, T, Cc, ACT[act], _stream())
    return out


def se_fwd(y, coef, W1, W2):
    """SELayer forward on the pre-BN tensor y (B,T,C) (models/QuartNetContextSE.py:8-23): the squeeze is taken through the BN
    coefficients coef [2][C] (mean_T(BN(y)) = a * mean_T(y) + b).  -> (ysum (B,C), pooled (B,C), hidden (B,C/8), scale (B,C)), f32."""
    B, T, Cc = y.shape
    dev = y.device
    ysum = torch.empty(B, Cc, dtype=torch.float32, device=dev)
    pooled = torch.empty(B, Cc, dtype=torch.float32, device=dev)
    hidden = torch.empty(B, Cc // 8, dtype=torch.float32, device=dev)
    scale = torch.empty(B, Cc, dtype=torch.float32, device=dev)
    call("lasr_seqsum", _p(y), _dt(y), B, T, Cc, _p(ysum), _stream())
    call("lasr_se_fwd", _p(ysum), _p(coef), _p(W1), _p(W2), B, T, Cc, _p(pooled), _p(hidden), _p(scale), _stream())
    return ysum, pooled, hidden, scale


def se_bwd(dout, y, coef, scale, hidden, pooled, W1, W2, y2=None, coef2=None, act: str = "relu"):
    """Backward of the SE branch of out = act(BN(y) * scale + BN2(y2)): -> (seg (B,C) = d(loss)/d(BN output) through the pooled
    mean, dW1 (C/8,C), dW2 (C,C/8))."""
    B, T, Cc = y.shape
    dev = y.device
    seg = torch.empty(B, Cc, dtype=torch.float32, device=dev)
    dW1 = torch.empty(Cc // 8, Cc, dtype=torch.float32, device=dev)
    dW2 = torch.empty(Cc, Cc // 8, dtype=torch.float32, device=dev)
    nb = int(_lib.load().lasr_se_bwd_workspace_bytes(B, Cc))
    ws = _ws(nb, dev)
    call("lasr_se_bwd", _p(dout), _p(y), _p(coef), _p(y2), _p(coef2), _p(scale), _p(hidden), _p(pooled), _p(W1), _p(W2), _dt(y), B, T, Cc,
         ACT[act], _p(seg), _p(dW1), _p(dW2), _p(ws), nb, _stream())
    return seg, dW1, dW2


def bn_act_bwd(dout, y, coef, saved, gamma, y2=None, coef2=None, saved2=None, gamma2=None, se_scale=None, se_grad=None,
               row_lens=None, act: str = "relu", fused: bool = False):
    """Returns (dy, dy2, dgamma, dbeta, dgamma2, dbeta2).  fused: pass 2 reduces pass 1's partial sums itself."""
    B, T, Cc = y.shape
    dev = y.device
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=dev)
    sums2 = torch.empty(2 * Cc, dtype=torch.float32, device=dev)
    nb = max(_lib.load().lasr_bn_bwd_workspace_bytes(B, T, Cc), _lib.load().lasr_bn_bwd_apply_workspace_bytes(Cc))
    ws = _ws(nb, dev)
    if fused:
        sums = sums2 = None
    call("lasr_bn_act_bwd_stats", _p(dout), _p(y), _p(coef), _p(saved), _p(y2), _p(coef2), _p(saved2), _p(se_scale),
         _p(se_grad), _p(sums), _p(sums2), _dt(y), B, T, Cc, ACT[act], _p(ws), nb, _stream())
    dy = torch.empty_like(y)
    dy2 = torch.empty_like(y) if y2 is not None else None
    dg, db = (torch.empty(Cc, dtype=torch.float32, device=dev) for _ in range(2))
    dg2, db2 = ((torch.empty(Cc, dtype=torch.float32, device=dev) for _ in range(2)) if y2 is not None else (None, None))
    call("lasr_bn_act_bwd_apply", _p(dout), _p(y), _p(coef), _p(saved), _p(gamma), _p(y2), _p(coef2), _p(saved2), _p(gamma2),
         _p(se_scale), _p(se_grad), _p(sums), _p(sums2), _p(row_lens), _p(dy), _p(dy2), _p(dg), _p(db), _p(dg2), _p(db2),
         _dt(y), B, T, Cc, ACT[act], _p(ws), ws.numel(), _stream())
    return dy, dy2, dg, db, dg2, db2


# ---------------------------------------------------------------------------------- head + loss
def log_softmax(logits: torch.Tensor, want_argmax: bool = True):
    shp = logits.shape
    Cc = shp[-1]
    N = logits.numel() // Cc
    out = torch.empty_like(logits)
    am = torch.empty(shp[:-1], dtype=torch.int32, device=logits.device) if want_argmax else None
    call("lasr_log_softmax", _p(logits), _p(out), _p(am), N, Cc, _stream())
    return out, am


def ctc_loss(logp: torch.Tensor, targets: torch.Tensor, in_lens: torch.Tensor, tgt_lens: torch.Tensor, blank: int,
             want_grad: bool = True, gscale: Optional[torch.Tensor] = None):
    """logp (B,T,C) f32 -> (nll (B), grad (B,T,C) | None); grad is torch's CTCLoss backward for grad_output=gscale."""
    B, T, Cc = logp.shape
    S = targets.shape[1] if targets.dim() == 2 else 0
    nll = torch.empty(B, dtype=torch.float32, device=logp.device)
    grad = torch.empty_like(logp) if want_grad else None
    nb = _lib.load().lasr_ctc_workspace_bytes(B, T, max(S, 1))
    ws = _ws(nb, logp.device)
    call("lasr_ctc_loss", _p(logp), _p(targets), _p(in_lens), _p(tgt_lens), B, T, Cc, max(S, 1), blank, _p(nll), _p(grad),
         _p(gscale), _p(ws), nb, _stream())
    return nll, grad


def greedy_decode(ids: torch.Tensor, lens: Optional[torch.Tensor], blank: int):
    B, T = ids.shape
    tokens = torch.empty(B, T, dtype=torch.int32, device=ids.device)
    n = torch.empty(B, dtype=torch.int32, device=ids.device)
    call("lasr_greedy_decode", _p(ids), _p(lens), B, T, blank, _p(tokens), _p(n), _stream())
    return tokens, n


# ---------------------------------------------------------------------------------- optimiser
def novograd_workspace(n_tensors: int, n_elems: int, device) -> torch.Tensor:
    """a ZEROED workspace a caller keeps across novograd_step(ws=...) calls (every call leaves it zeroed)"""
    return torch.zeros(max(int(_lib.load().lasr_novograd_workspace_bytes(n_tensors, n_elems)), 256), dtype=torch.uint8, device=device)


def novograd_step(params, grads, exp_avg, exp_avg_sq, offsets, lr_dev, beta1=0.8, beta2=0.5, eps=1e-8, weight_decay=0.0,
                  grad_scale=1.0, ws: Optional[torch.Tensor] = None):
    """ws: a workspace from novograd_workspace() kept by the caller - the step then issues no memset of its own"""
    n_t = exp_avg_sq.numel()
    n = params.numel()
    nb = _lib.load().lasr_novograd_workspace_bytes(n_t, n)
    if ws is not None:
        call("lasr_novograd_step_keep", _p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), _p(offsets), n_t, n, _p(lr_dev), beta1,
             beta2, eps, weight_decay, grad_scale, _p(ws), ws.numel(), _stream())
        return
    ws = _ws(nb, params.device)
    call("lasr_novograd_step", _p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), _p(offsets), n_t, n, _p(lr_dev), beta1,
         beta2, eps, weight_decay, grad_scale, _p(ws), nb, _stream())


def edit_distance_batch(tokens: torch.Tensor, n_tokens: torch.Tensor, targets: torch.Tensor, target_lens: torch.Tensor,
                        space_id: int = -1, totals: Optional[torch.Tensor] = None):
    """Levenshtein distance of every utterance on the device: tokens (B,T) i32 + n_tokens (B) i32 from greedy_decode,
    targets (B,S) i64 + target_lens (B) i32.  space_id < 0: per token (CER); >= 0: per word.  Returns (dist (B) i32,
    ref_units (B) i32); ``totals`` (2) i64, if given, is incremented by their sums (no host synchronisation)."""
    B = tokens.shape[0]
    dev = tokens.device
    dist = torch.empty(B, dtype=torch.int32, device=dev)
    units = torch.empty(B, dtype=torch.int32, device=dev)
    call("lasr_edit_distance_batch", _p(tokens), _p(n_tokens), tokens.shape[1], _p(targets), _p(target_lens), targets.shape[1], B,
         int(space_id), _p(dist), _p(units), _p(totals), _stream())
    return dist, units


def step_metrics(loss: torch.Tensor, dist: torch.Tensor, units: torch.Tensor, acc: torch.Tensor) -> None:
    """fold one step's loss and batch WER into the device accumulators acc (7 f64): see lasr_step_metrics"""
    if acc.dtype != torch.float64 or acc.numel() < 7:
        raise TypeError("acc must hold 7 float64 values")
    call("lasr_step_metrics", _p(loss), _p(dist), _p(units), dist.numel(), _p(acc), _stream())
