"""CPU oracle for the QuartzNet-CTC training path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, plain-torch/numpy (CPU, fp32) restatement of what
kouyt5/lightning-asr computes on its training hot path.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``lightning_asr_amd/`` imports it, and the
product path has no CPU fallback.

Pinning (see DESIGN.md "Oracle"):
  * model / NovoGrad / LR schedule: pinned against the reference itself, imported
    from /root/reference in the dev container by ``oracle/make_golden.py`` which
    writes ``tests/golden/*.npz`` (outputs only; weights are formula-generated).
  * CTC: ``torch.nn.functional.ctc_loss`` (what ``train.py:196`` calls) plus an
    independent numpy alpha/beta restatement below for small cases.
  * mel front-end: torchaudio 0.8.1 is not installable here, so the mel part is
    "parity unpinned" by any executable reference; it restates the published
    torchaudio 0.8.1 algorithm (Spectrogram/MelScale/AmplitudeToDB) with the
    argument values at ``data_module.py:68-71``.

Reference citations are ``path:line`` under /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Mel front-end  (data_module.py:59-73 constants, :150-174 parse_audio)
# --------------------------------------------------------------------------------------
SR = 16000
N_FFT = 512
WIN = 320          # int(0.02 * 16000)            data_module.py:66
HOP = 160          # win // 2                     data_module.py:67
PAD = 32           # MelSpectrogram(pad=32)       data_module.py:68
N_MELS = 64
N_FREQ = N_FFT // 2 + 1
PREEMPH = 0.97     # data_module.py:157
DITHER = 1e-5      # data_module.py:155
AMIN = 1e-10       # torchaudio AmplitudeToDB default amin


def hann_window_padded() -> torch.Tensor:
    """Periodic Hann(320) centred in a 512 frame (torch.stft pads the window
    (n_fft - win_length)//2 = 96 zeros on the left, the rest on the right)."""
    n = torch.arange(WIN, dtype=torch.float64)
    w = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / WIN)
    out = torch.zeros(N_FFT, dtype=torch.float64)
    left = (N_FFT - WIN) // 2
    out[left:left + WIN] = w
    return out.float()


def mel_filterbank() -> torch.Tensor:
    """(257, 64) HTK triangular filterbank, torchaudio 0.8.1 create_fb_matrix with
    f_min=0, f_max=sr//2, norm=None (computed in f32 as torchaudio does)."""
    all_freqs = torch.linspace(0, SR // 2, N_FREQ)
    m_min = 2595.0 * math.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * math.log10(1.0 + (SR // 2) / 700.0)
    m_pts = torch.linspace(m_min, m_max, N_MELS + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)       # (257, 66)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def num_frames(n_samples: int) -> int:
    return 1 + (n_samples + 2 * PAD) // HOP


def preemphasis(y: torch.Tensor) -> torch.Tensor:
    """y: (1, L).  Keeps y[0] (data_module.py:157)."""
    return torch.cat((y[:, :1], y[:, 1:] - PREEMPH * y[:, :-1]), dim=1)


def mel_power(y: torch.Tensor) -> torch.Tensor:
    """y: (1, L) pre-emphasised wave -> (1, 64, T) power mel spectrogram (in y's dtype: f32 is
    the reference arithmetic, f64 is used by tests to price f32 FFT round-off)."""
    y = F.pad(y, (PAD, PAD), "constant")
    spec = torch.stft(y, N_FFT, hop_length=HOP, win_length=N_FFT, window=hann_window_padded().to(y.dtype),
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                    # |.|^2, (1, 257, T)
    fb = mel_filterbank().to(y.dtype)
    return torch.matmul(power.transpose(1, 2), fb).transpose(1, 2)


def amplitude_to_db(x: torch.Tensor) -> torch.Tensor:
    """AmplitudeToDB('power'), top_db=None, ref=1 (data_module.py:71)."""
    return 10.0 * torch.log10(torch.clamp(x, min=AMIN))


def normalize_utt(y: torch.Tensor) -> torch.Tensor:
    """(y - mean) / std with the unbiased std over all 64*T values (data_module.py:171-172)."""
    std, mean = torch.std_mean(y)
    return (y - mean) / std


def spec_augment_draw(rng, n_freq: int, n_time: int, freq_mask=27, time_mask=0.07):
    """Draw (rect_x, w_x, rect_y, w_y) exactly as data_module.py:97-122 does from a
    ``random.Random``-like ``rng`` (4 uniform() calls, in this order)."""
    if isinstance(freq_mask, float):
        freq_mask = int(n_freq * freq_mask)
    if isinstance(time_mask, float):
        time_mask = int(n_time * time_mask)
    w_x = int(rng.uniform(0, freq_mask))
    w_y = int(rng.uniform(0, time_mask))
    rect_x = int(rng.uniform(0, n_freq - w_x))
    rect_y = int(rng.uniform(0, n_time - w_y))
    return rect_x, w_x, rect_y, w_y


def spec_augment_apply(x: torch.Tensor, rect_x, w_x, rect_y, w_y) -> torch.Tensor:
    """Zero rows [rect_x, rect_x+w_x) and columns [rect_y, rect_y+w_y) of (1, 64, T)."""
    x = x.clone()
    x[0, rect_x:rect_x + w_x, :] = 0
    x[0, :, rect_y:rect_y + w_y] = 0
    return x


def sub_sequence(y: torch.Tensor, u_len: float, u_loc: float, weight: float = 0.98) -> torch.Tensor:
    """Bug-compatible ``sub_secquence`` (data_module.py:138-148): the slice END is
    ``target_length`` (not location+target_length).  u_len/u_loc are the two
    uniform(0,1) draws, mapped as np.random.uniform(a,b) = a + (b-a)*u."""
    length = y.shape[1]
    target_length = int(length * (weight + (1.0 - weight) * u_len))
    location = int((length - target_length) * u_loc)
    return y[:, location:target_length]


def parse_wave(y: torch.Tensor, dither: Optional[torch.Tensor] = None,
               aug: Optional[Tuple[int, int, int, int]] = None, normalize: bool = True,
               crop: Optional[Tuple[float, float]] = None) -> torch.Tensor:
    """(1, L) wave -> (1, 64, T) normalised log-mel, the chain of data_module.py:150-174
    with the random dither passed in explicitly (None = dither off).  crop = (u_len, u_loc): the two
    uniforms of the training-time ``sub_secquence``, applied where the reference applies it - AFTER
    dither and pre-emphasis (:155-159), so the crop's first sample is y[loc] - 0.97 y[loc-1]."""
    if y.dtype != torch.float64:
        y = y.float()
    if dither is not None:
        y = y + DITHER * dither.to(y.dtype)
    y = preemphasis(y)
    if crop is not None:
        y = sub_sequence(y, crop[0], crop[1])
    db = amplitude_to_db(mel_power(y))
    if aug is not None:
        db = spec_augment_apply(db, *aug)
    return normalize_utt(db) if normalize else db


def collate(feats: Sequence[torch.Tensor], texts: Sequence[Sequence[int]]):
    """Pad-to-longest collate (data_module.py:222-248) -> (inputs, targets, pct, target_sizes)."""
    B = len(feats)
    t_max = max(f.size(2) for f in feats)
    s_max = max(len(t) for t in texts)
    inputs = torch.zeros(B, 1, feats[0].size(1), t_max)
    pct = torch.zeros(B, dtype=torch.float32)
    tsz = torch.zeros(B, dtype=torch.int32)
    targets = torch.zeros(B, s_max, dtype=torch.int64)
    for i, (f, t) in enumerate(zip(feats, texts)):
        T = f.size(2)
        inputs[i, 0, :, :T] = f[0]
        pct[i] = T / float(t_max)
        tsz[i] = len(t)
        targets[i, :len(t)] = torch.tensor(list(t), dtype=torch.int64)
    return inputs, targets, pct, tsz


# --------------------------------------------------------------------------------------
# Model  (models/QuartNet.py, models/QuartNetContext.py, models/QuartNetContextSE.py)
# --------------------------------------------------------------------------------------
BN_EPS = 1e-3      # nn.BatchNorm1d(out_ch, eps=1e-3)    models/QuartNet.py:24
BN_MOM = 0.1
SE_REDUCTION = 8   # models/QuartNetContextSE.py:46
LSTM_HIDDEN = 40   # models/QuartNetContext.py:157

VARIANTS = ("plain", "context", "context_se")


def block_table(variant: str) -> List[Tuple[str, int, int, int]]:
    """[(name, in_ch, out_ch, k)] of the residual blocks, in forward order."""
    ctx = variant != "plain"
    t = [("block1", 256, 256, 33), ("block12", 256, 256, 33), ("block13", 256, 256, 33),
         ("block2", 256, 256, 39), ("block22", 256, 256, 39), ("block23", 256, 256, 39),
         ("block3", 336 if ctx else 256, 512, 51), ("block32", 512, 512, 51), ("block33", 512, 512, 51),
         ("block4", 512, 512, 63), ("block42", 512, 512, 63), ("block43", 512, 512, 63),
         ("block5", 512, 512, 75)]
    if ctx:
        t.append(("block6", 512, 512, 87))
    return t


def _sep_shapes(prefix: str, ci: int, co: int, k: int, se: bool):
    out = [(prefix + ".depthwise_conv.weight", (ci, 1, k)),
           (prefix + ".pointwise_conv.weight", (co, ci, 1))]
    out += _bn_shapes(prefix + ".bn", co)
    if se:
        out += [(prefix + ".se.fc.0.weight", (co // SE_REDUCTION, co)),
                (prefix + ".se.fc.2.weight", (co, co // SE_REDUCTION))]
    return out


def _bn_shapes(prefix: str, c: int):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)),
            (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,)),
            (prefix + ".num_batches_tracked", ())]


def state_shapes(variant: str, n_class: int, in_c: int = 64) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) list == the reference MyModel2.state_dict() layout."""
    se = variant == "context_se"
    out = _sep_shapes("encoder.first_cnn", in_c, 256, 33, se)
    for name, ci, co, k in block_table(variant):
        p = "encoder." + name
        out += [(p + ".reside.0.weight", (co, ci, 1))] + _bn_shapes(p + ".reside.1", co)
        out += _sep_shapes(p + ".seq.0", ci, co, k, se)
    out += [("encoder.last_cnn2.0.weight", (1024, 512, 1))] + _bn_shapes("encoder.last_cnn2.1", 1024)
    if variant != "plain":
        r = "encoder.context_rnn.rnn."
        for sfx in ("", "_reverse"):
            out += [(r + "weight_ih_l0" + sfx, (4 * LSTM_HIDDEN, 256)),
                    (r + "weight_hh_l0" + sfx, (4 * LSTM_HIDDEN, LSTM_HIDDEN)),
                    (r + "bias_ih_l0" + sfx, (4 * LSTM_HIDDEN,)),
                    (r + "bias_hh_l0" + sfx, (4 * LSTM_HIDDEN,))]
    out += [("decoder.weight", (n_class, 1024, 1)), ("decoder.bias", (n_class,))]
    return out


def is_buffer(key: str) -> bool:
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def hash_uniform(n: int, stream: int) -> np.ndarray:
    """n float64 in [0,1): splitmix64 finaliser of (i + stream*golden) — exact integer
    arithmetic, so identical on every machine / library version."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def formula_state(variant: str, n_class: int, in_c: int = 64, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic formula-defined weights (no blobs in the repo; SURVEY §8c fixture 1).
    Tensor #idx, element #i:  a*(2u-1) with u = hash_uniform(i, idx), a = gain/sqrt(fan_in)
    (the PyTorch default-init range, which keeps the BN stack well conditioned: measured
    noise amplification input->log-probs ~10x vs ~1000x for a sine pattern);
    BN gamma = 1 + 0.1(2u-1), beta = 0.1(2u-1); running_mean 0, running_var 1."""
    sd: Dict[str, torch.Tensor] = {}
    for idx, (key, shape) in enumerate(state_shapes(variant, n_class, in_c)):
        n = int(np.prod(shape)) if len(shape) else 1
        if key.endswith("num_batches_tracked"):
            t = torch.zeros((), dtype=torch.int64)
        elif key.endswith("running_mean"):
            t = torch.zeros(shape)
        elif key.endswith("running_var"):
            t = torch.ones(shape)
        else:
            u = torch.from_numpy(2.0 * hash_uniform(n, idx) - 1.0)
            if ".bn." in key or ".reside.1." in key or ".last_cnn2.1." in key:
                t = ((1.0 if key.endswith("weight") else 0.0) + 0.1 * u).float().reshape(shape)
            else:
                if "rnn" in key:
                    fan_in = LSTM_HIDDEN
                elif key == "decoder.bias":
                    fan_in = 1024
                else:
                    fan_in = int(np.prod(shape[1:]))
                t = (gain / math.sqrt(fan_in) * u).float().reshape(shape)
        sd[key] = t
    return sd


def random_state(variant: str, n_class: int, seed: int = 0, in_c: int = 64) -> Dict[str, torch.Tensor]:
    """PyTorch-default-like random init (kaiming-uniform a=sqrt(5) => U(-1/sqrt(fan_in), ..))."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in state_shapes(variant, n_class, in_c):
        if key.endswith("num_batches_tracked"):
            t = torch.zeros((), dtype=torch.int64)
        elif key.endswith("running_mean"):
            t = torch.zeros(shape)
        elif key.endswith("running_var"):
            t = torch.ones(shape)
        elif (".bn." in key or ".reside.1." in key or ".last_cnn2.1." in key):
            t = torch.ones(shape) if key.endswith("weight") else torch.zeros(shape)
        else:
            if "rnn" in key:
                bound = 1.0 / math.sqrt(LSTM_HIDDEN)
            elif key == "decoder.bias":
                bound = 1.0 / math.sqrt(1024)
            else:
                bound = 1.0 / math.sqrt(int(np.prod(shape[1:])))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        sd[key] = t
    return sd


def mask_lengths(T: int, pct: torch.Tensor) -> torch.Tensor:
    """int32 trunc of the f32 product T*pct (models/QuartNet.py:311, train.py:76)."""
    return torch.mul(T, pct.float()).int()


def _time_mask(x: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    T = x.size(2)
    keep = (torch.arange(T, device=x.device).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1)
    return x * keep.to(x.dtype)


def activation(x: torch.Tensor, act: str) -> torch.Tensor:
    if act == "relu":
        return F.relu(x)
    if act == "swish":          # activate_fun/Swish.py:9-10
        return x * torch.sigmoid(x)
    raise ValueError(act)


class OracleModel:
    """Functional restatement of MyModel2 (three variants) over a flat ``state`` dict whose
    keys equal the reference state_dict keys.  ``training`` selects batch-stat BN and
    updates the running buffers in ``state`` in place, like nn.BatchNorm1d."""

    def __init__(self, variant: str, n_class: int, mask: bool = True, act: str = "relu",
                 state: Optional[Dict[str, torch.Tensor]] = None, in_c: int = 64):
        assert variant in VARIANTS
        self.variant, self.n_class, self.mask, self.act, self.in_c = variant, n_class, mask, act, in_c
        self.state = state if state is not None else random_state(variant, n_class, 0, in_c)
        self.training = True
        self.taps: Dict[str, torch.Tensor] = {}     # per-layer activations of the last forward
        self.keep_taps = False

    # -- parameter handling ------------------------------------------------------------
    def param_keys(self) -> List[str]:
        return [k for k in self.state if not is_buffer(k)]

    def parameters(self) -> List[torch.Tensor]:
        return [self.state[k] for k in self.param_keys()]

    def requires_grad_(self, flag: bool = True):
        for k in self.param_keys():
            self.state[k].requires_grad_(flag)
        return self

    # -- layers --------------------------------------------------------------------------
    def _bn(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        s = self.state
        if self.training:
            s[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, s[prefix + ".running_mean"], s[prefix + ".running_var"],
                            s[prefix + ".weight"], s[prefix + ".bias"], self.training, BN_MOM, BN_EPS)

    def _sep(self, x, lens, prefix: str, k: int, stride: int, last: bool) -> torch.Tensor:
        """SeprationConv (models/QuartNet.py:29-39): dw -> pw -> mask -> BN -> [SE] -> act."""
        s = self.state
        ci = x.size(1)
        u = F.conv1d(x, s[prefix + ".depthwise_conv.weight"], None, stride, k // 2, 1, ci)
        y = F.conv1d(u, s[prefix + ".pointwise_conv.weight"])
        if self.mask:
            # MaskCNN: lens = int32(T'*pct) of this tensor's own T' (models/QuartNet.py:311)
            y = _time_mask(y, lens)
        z = self._bn(y, prefix + ".bn")
        if self.variant == "context_se":
            pooled = z.mean(dim=2)                                        # over ALL T' incl. padding
            g = torch.sigmoid(F.linear(F.relu(F.linear(pooled, s[prefix + ".se.fc.0.weight"])),
                                       s[prefix + ".se.fc.2.weight"]))
            z = z * g.unsqueeze(2)
        if not last:
            z = activation(z, self.act)
        return z

    def _block(self, x, lens, name: str, k: int) -> torch.Tensor:
        """QuartNetBlock, repeat=1 (models/QuartNet.py:71-78)."""
        p = "encoder." + name
        main = self._sep(x, lens, p + ".seq.0", k, 1, True)
        res = self._bn(F.conv1d(x, self.state[p + ".reside.0.weight"]), p + ".reside.1")
        return activation(main + res, self.act)

    def _bilstm(self, x: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        """pack_padded -> BiLSTM(256->40) -> pad_packed (models/QuartNetContext.py:186-199):
        gates i,f,g,o; reverse direction starts at each sample's own last valid frame;
        outputs are zero for t >= len_b.  x: (B, T, 256) -> (B, T, 80)."""
        s = self.state
        B, T, _ = x.shape
        H = LSTM_HIDDEN
        outs = []
        r = "encoder.context_rnn.rnn."
        valid = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).to(x.dtype)   # (B, T)
        for sfx, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
            w_ih, w_hh = s[r + "weight_ih_l0" + sfx], s[r + "weight_hh_l0" + sfx]
            bias = s[r + "bias_ih_l0" + sfx] + s[r + "bias_hh_l0" + sfx]
            gx = F.linear(x, w_ih, bias)                                         # (B, T, 4H)
            h = x.new_zeros(B, H)
            c = x.new_zeros(B, H)
            out = [None] * T
            for t in order:
                g = gx[:, t] + F.linear(h, w_hh)
                i, f, gg, o = g.split(H, dim=1)
                c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h_new = torch.sigmoid(o) * torch.tanh(c_new)
                m = valid[:, t:t + 1]
                c = m * c_new + (1 - m) * c
                h = m * h_new + (1 - m) * h
                out[t] = m * h_new
            outs.append(torch.stack(out, dim=1))
        return torch.cat(outs, dim=2)

    # -- forward -------------------------------------------------------------------------
    def encode(self, inputs: torch.Tensor, pct: torch.Tensor) -> torch.Tensor:
        x = inputs.squeeze(1)
        T1 = (x.size(2) + 2 * 16 - 33) // 2 + 1
        lens = mask_lengths(T1, pct)
        tap = self.taps.__setitem__ if self.keep_taps else (lambda k, v: None)
        x = self._sep(x, lens, "encoder.first_cnn", 33, 2, False)
        tap("first_cnn", x)
        for name, ci, co, k in block_table(self.variant):
            if name == "block3" and self.variant != "plain":
                assert int(lens.max()) == x.size(2), "reference torch.cat needs max(len)==T'"
                ctx = self._bilstm(x.transpose(1, 2), lens)
                tap("context", ctx)
                x = torch.cat((x, ctx.transpose(1, 2)), dim=1)
            x = self._block(x, lens, name, k)
            tap(name, x)
        y = F.conv1d(x, self.state["encoder.last_cnn2.0.weight"])
        x = activation(self._bn(y, "encoder.last_cnn2.1"), self.act)
        tap("last_cnn2", x)
        return x

    def forward(self, inputs: torch.Tensor, pct: torch.Tensor) -> torch.Tensor:
        """inputs (B,1,F,T) f32, pct (B,) -> log-probs (B, T', C)  (models/QuartNet.py:280-291)."""
        x = self.encode(inputs, pct)
        logits = F.conv1d(x, self.state["decoder.weight"], self.state["decoder.bias"])
        if self.keep_taps:
            self.taps["logits"] = logits
        return F.log_softmax(logits.transpose(1, 2), dim=-1)

    __call__ = forward


# --------------------------------------------------------------------------------------
# CTC  (train.py:76-78,196)
# --------------------------------------------------------------------------------------
def ctc_loss_per_sample(log_probs_btc: torch.Tensor, targets: torch.Tensor, t_lengths: torch.Tensor,
                        target_lengths: torch.Tensor, blank: int) -> torch.Tensor:
    """reduction='none' CTC on (B,T,C) log-probs, zero_infinity=False.  Returns (B,) NLL."""
    return F.ctc_loss(log_probs_btc.transpose(0, 1), targets, t_lengths.int(), target_lengths.int(),
                      blank=blank, reduction="none", zero_infinity=False)


def training_loss(log_probs_btc, targets, pct, target_lengths, blank) -> torch.Tensor:
    """mean_b CTC, lengths = int(T'*pct) (train.py:76-78)."""
    tl = mask_lengths(log_probs_btc.size(1), pct)
    return torch.mean(ctc_loss_per_sample(log_probs_btc, targets, tl, target_lengths, blank))


def ctc_numpy(log_probs_tc: np.ndarray, target: Sequence[int], blank: int):
    """Independent float64 log-space alpha/beta CTC for ONE sample (small cases only).
    log_probs_tc: (T, C) already truncated to the input length.  Returns (nll, grad wrt
    log_probs (T, C)) with the torch convention grad = -(occupancy)  (i.e. d nll / d log_prob)."""
    T, C = log_probs_tc.shape
    lp = log_probs_tc.astype(np.float64)
    ext = [blank]
    for s in target:
        ext += [int(s), blank]
    S = len(ext)
    NEG = -np.inf

    def lse(*a):
        m = max(a)
        if m == NEG:
            return NEG
        return m + math.log(sum(math.exp(x - m) for x in a))

    alpha = np.full((T, S), NEG)
    beta = np.full((T, S), NEG)
    if T == 0:
        return (0.0 if S == 1 else np.inf), np.zeros_like(lp)
    alpha[0, 0] = lp[0, ext[0]]
    if S > 1:
        alpha[0, 1] = lp[0, ext[1]]
    for t in range(1, T):
        for s in range(S):
            a = [alpha[t - 1, s]]
            if s >= 1:
                a.append(alpha[t - 1, s - 1])
            if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                a.append(alpha[t - 1, s - 2])
            alpha[t, s] = lse(*a) + lp[t, ext[s]]
    ll = lse(alpha[T - 1, S - 1], alpha[T - 1, S - 2]) if S > 1 else alpha[T - 1, 0]
    beta[T - 1, S - 1] = lp[T - 1, ext[S - 1]]
    if S > 1:
        beta[T - 1, S - 2] = lp[T - 1, ext[S - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(S):
            b = [beta[t + 1, s]]
            if s + 1 < S:
                b.append(beta[t + 1, s + 1])
            if s + 2 < S and ext[s + 2] != blank and ext[s + 2] != ext[s]:
                b.append(beta[t + 1, s + 2])
            beta[t, s] = lse(*b) + lp[t, ext[s]]
    grad = np.zeros_like(lp)
    if ll == NEG:
        return np.inf, grad
    occ = np.full((T, C), NEG)
    for t in range(T):
        for s in range(S):
            v = alpha[t, s] + beta[t, s]
            occ[t, ext[s]] = lse(occ[t, ext[s]], v)
    with np.errstate(over="ignore"):
        grad = -np.exp(occ - ll - lp)
    return -ll, grad


# --------------------------------------------------------------------------------------
# Greedy decode + WER  (utils/asr_metrics.py:138-228)
# --------------------------------------------------------------------------------------
def greedy_collapse(ids: Sequence[int], blank: int) -> List[int]:
    out, prev = [], blank
    for p in ids:
        if (p != prev or prev == blank) and p != blank:
            out.append(int(p))
        prev = p
    return out


def greedy_decode(argmax_bt: torch.Tensor, lengths: Optional[torch.Tensor], labels: Sequence[str]) -> List[str]:
    blank = len(labels)
    hyps = []
    for b in range(argmax_bt.size(0)):
        ids = argmax_bt[b].tolist()
        if lengths is not None:
            ids = ids[: int(lengths[b])]
        hyps.append("".join(labels[c] for c in greedy_collapse(ids, blank)))
    return hyps


def levenshtein(a: Sequence, b: Sequence) -> int:
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def word_error_rate(hyps: Sequence[str], refs: Sequence[str], use_cer: bool = False) -> float:
    """utils/asr_metrics.py:26-59 — sum of edit distances / sum of reference lengths."""
    scores = words = 0
    for h, r in zip(hyps, refs):
        hl, rl = (list(h), list(r)) if use_cer else (h.split(), r.split())
        words += len(rl)
        scores += levenshtein(hl, rl)
    return scores / words if words else float("inf")


# --------------------------------------------------------------------------------------
# NovoGrad + LR schedule  (scheduler/novograd.py:75-145, scheduler/cosine_annearing_with_warmup.py)
# --------------------------------------------------------------------------------------
class NovogradState:
    def __init__(self, n: int):
        self.exp_avg: List[Optional[torch.Tensor]] = [None] * n
        self.exp_avg_sq: List[Optional[torch.Tensor]] = [None] * n


@torch.no_grad()
def novograd_step(params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], st: NovogradState,
                  lr: float, beta1: float = 0.8, beta2: float = 0.5, eps: float = 1e-8,
                  weight_decay: float = 0.0) -> None:
    """One step of the reference configuration (betas=(0.8,0.5), no grad_averaging/amsgrad/luc;
    train.py:46).  ``v`` is the per-tensor scalar ||g||^2 EMA, initialised to the first norm
    ("if exp_avg_sq == 0: copy", scheduler/novograd.py:115-118)."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if st.exp_avg[i] is None:
            st.exp_avg[i] = torch.zeros_like(p)
            st.exp_avg_sq[i] = torch.zeros((), dtype=p.dtype, device=p.device)
        norm = g.norm().pow(2)
        v = st.exp_avg_sq[i]
        if v == 0:
            v.copy_(norm)
        else:
            v.mul_(beta2).add_(norm, alpha=1.0 - beta2)
        u = g.clone().div_(v.sqrt().add_(eps))
        if weight_decay != 0:
            u.add_(p, alpha=weight_decay)
        st.exp_avg[i].mul_(beta1).add_(u)
        p.add_(st.exp_avg[i], alpha=-lr)


class CosineWarmupRestarts:
    """Host restatement of CosineAnnealingWarmupRestarts driven with step() once per batch
    (scheduler/cosine_annearing_with_warmup.py:47-89, epoch=None branch; args train.py:53-55).
    ``lr`` right after construction is ``min_lr`` (init_lr), as is the value after the
    _LRScheduler constructor's implicit first step()."""

    def __init__(self, first_cycle_steps: int, cycle_mult: float = 1.0, max_lr: float = 0.1,
                 min_lr: float = 0.001, warmup_steps: int = 0, gamma: float = 1.0):
        assert warmup_steps < first_cycle_steps
        self.first_cycle_steps, self.cycle_mult = first_cycle_steps, cycle_mult
        self.base_max_lr = self.max_lr = max_lr
        self.min_lr, self.warmup_steps, self.gamma = min_lr, warmup_steps, gamma
        self.cur_cycle_steps = first_cycle_steps
        self.cycle = 0
        self.step_in_cycle = -1
        self.last_epoch = -1
        self.lr = min_lr
        self.step()                      # _LRScheduler.__init__ calls step() once

    def _value(self) -> float:
        if self.step_in_cycle == -1:
            return self.min_lr
        if self.step_in_cycle < self.warmup_steps:
            return (self.max_lr - self.min_lr) * self.step_in_cycle / self.warmup_steps + self.min_lr
        return self.min_lr + (self.max_lr - self.min_lr) * (
            1 + math.cos(math.pi * (self.step_in_cycle - self.warmup_steps)
                         / (self.cur_cycle_steps - self.warmup_steps))) / 2

    def step(self) -> float:
        self.last_epoch += 1
        self.step_in_cycle += 1
        if self.step_in_cycle >= self.cur_cycle_steps:
            self.cycle += 1
            self.step_in_cycle -= self.cur_cycle_steps
            self.cur_cycle_steps = int((self.cur_cycle_steps - self.warmup_steps) * self.cycle_mult) + self.warmup_steps
        self.max_lr = self.base_max_lr * (self.gamma ** self.cycle)
        self.lr = self._value()
        return self.lr


# --------------------------------------------------------------------------------------
# Whole training step (train.py:64-86 + optimiser), used by tests and bench cpu_baseline
# --------------------------------------------------------------------------------------
def train_step(model: OracleModel, opt: NovogradState, inputs, targets, pct, target_sizes, lr: float,
               weight_decay: float = 1e-3) -> Tuple[float, List[torch.Tensor]]:
    """fwd + CTC + bwd + NovoGrad on CPU.  Returns (loss, grads)."""
    model.training = True
    model.requires_grad_(True)
    params = model.parameters()
    for p in params:
        p.grad = None
    lp = model.forward(inputs, pct)
    loss = training_loss(lp, targets, pct, target_sizes, blank=model.n_class - 1)
    loss.backward()
    grads = [p.grad.detach().clone() for p in params]
    novograd_step([p.data for p in params], [g.clone() for g in grads], opt, lr, 0.8, 0.5, 1e-8, weight_decay)
    return float(loss.detach()), grads


def synth_batch(B: int, n_samples: int, S: int, n_vocab: int, seed: int = 1234):
    """SURVEY §8d synthetic batch: wave = 0.1*N(0,1), targets ~U{0..V-1} with no adjacent
    repeats (keeps CTC feasible for any T' >= S, note N9)."""
    g = torch.Generator().manual_seed(seed)
    wave = 0.1 * torch.randn(B, n_samples, generator=g)
    tg = torch.randint(0, n_vocab, (B, S), generator=g)
    for b in range(B):
        for s in range(1, S):
            if tg[b, s] == tg[b, s - 1]:
                tg[b, s] = (tg[b, s] + 1) % n_vocab
    return wave, tg.long(), torch.full((B,), S, dtype=torch.int32)
