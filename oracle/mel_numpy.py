"""TEST INFRASTRUCTURE - a SECOND, independent restatement of the reference's log-mel front-end, in numpy f64.

Why it exists: torchaudio 0.8.1 (`/root/reference/requirements.txt:4`) is neither in the reference tree nor installable here and the
reference holds no feature vectors, so the front-end's parity is UNPINNED by any executable reference.  The first
restatement (`oracle/ref_cpu.py::mel_power` and friends) leans on `torch.stft`; this one is written from the documented algorithm of
torchaudio 0.8.1's `MelSpectrogram` / `Spectrogram` / `MelScale` / `AmplitudeToDB` with the arguments of
`/root/reference/data_module.py:68-71` and the chain of `:150-174`, and shares NOTHING with the first: frames are cut by hand with
explicit reflect indices, the transform is `numpy.fft.rfft`, every triangular filter is evaluated from its three corner frequencies.
`tests/test_oracle_golden.py` asserts the two agree to f64 round-off - that pins nothing to the reference, but a framing / window-
centring / filter-edge slip can no longer hide inside one implementation.

Only tests/ may import this (like everything under oracle/).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

SR = 16000          # data_module.py:59 (sr)
N_FFT = 512         # data_module.py:68
WIN = 320           # int(0.02 * 16000), data_module.py:66
HOP = 160           # WIN // 2, data_module.py:67
PAD = 32            # MelSpectrogram(pad=32), data_module.py:68
N_MELS = 64


def hann_periodic(n: int) -> np.ndarray:
    """torch.hann_window(n) (periodic=True): w[i] = sin^2(pi i / n)"""
    i = np.arange(n, dtype=np.float64)
    return np.sin(math.pi * i / n) ** 2


def frame_matrix(y: np.ndarray) -> np.ndarray:
    """Spectrogram(pad=32, n_fft=512, hop=160, center=True, pad_mode='reflect') framing of a 1-D signal, by index arithmetic.
    The signal is first extended by PAD zeros on both sides (`Spectrogram.pad`), then reflected by n_fft // 2 without repeating the
    edge sample; frame t covers positions [t*hop - n_fft//2, t*hop + n_fft//2) of the zero-extended signal."""
    L = y.shape[0] + 2 * PAD
    half = N_FFT // 2
    if L <= half:
        raise ValueError("reflect padding needs more than n_fft // 2 samples (torch.stft raises as well)")
    T = 1 + L // HOP
    pos = np.arange(T)[:, None] * HOP - half + np.arange(N_FFT)[None, :]          # position in the zero-extended signal
    pos = np.where(pos < 0, -pos, pos)                                            # reflect at the left edge (sample 0 not repeated)
    pos = np.where(pos > L - 1, 2 * (L - 1) - pos, pos)                           # and at the right edge
    z = np.concatenate([np.zeros(PAD), y.astype(np.float64), np.zeros(PAD)])
    return z[pos]


def power_spectrum(y: np.ndarray, window: Optional[np.ndarray] = None) -> np.ndarray:
    """(T, 257) |STFT|^2 with the 320-sample periodic Hann window centred in the 512-sample frame (torch.stft pads a short window
    with (n_fft - win_length) // 2 zeros on the left), normalized=False, onesided.  window: another 320-sample table (the first
    restatement's f32-rounded one, to compare the two at f64 round-off instead of at the tables' f32 round-off)"""
    win = np.zeros(N_FFT)
    left = (N_FFT - WIN) // 2
    win[left:left + WIN] = hann_periodic(WIN) if window is None else window
    spec = np.fft.rfft(frame_matrix(y) * win[None, :], axis=1)
    return spec.real ** 2 + spec.imag ** 2


def hz_to_mel(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)          # HTK (torchaudio 0.8.1 has no other scale)


def mel_to_hz(m: float) -> float:
    return 700.0 * (10.0 ** (m / 2595.0) - 1.0)


def filterbank() -> np.ndarray:
    """(257, 64) triangular filters, create_fb_matrix(n_freqs=257, f_min=0, f_max=sr // 2, n_mels=64, norm=None): filter m rises from
    corner m to corner m+1 and falls to corner m+2, corners equally spaced on the mel axis; bins at linspace(0, sr // 2, 257)."""
    f_max = float(SR // 2)
    corners = [mel_to_hz(hz_to_mel(0.0) + (hz_to_mel(f_max) - hz_to_mel(0.0)) * k / (N_MELS + 1)) for k in range(N_MELS + 2)]
    fb = np.zeros((N_FFT // 2 + 1, N_MELS))
    for m in range(N_MELS):
        lo, mid, hi = corners[m], corners[m + 1], corners[m + 2]
        for k in range(N_FFT // 2 + 1):
            f = f_max * k / (N_FFT // 2)
            fb[k, m] = max(0.0, min((f - lo) / (mid - lo), (hi - f) / (hi - mid)))
    return fb


def parse_wave(y: np.ndarray, dither: Optional[np.ndarray] = None, aug: Optional[Tuple[int, int, int, int]] = None,
               normalize: bool = True, crop: Optional[Tuple[float, float]] = None, window: Optional[np.ndarray] = None,
               fb: Optional[np.ndarray] = None) -> np.ndarray:
    """AudioParser.parse_audio after the file read (data_module.py:155-172) on a 1-D signal -> (64, T):
    y += 1e-5 noise; y[i] -= 0.97 y[i-1] for i >= 1; [training crop y[loc:target]]; mel power; 10 log10(max(., 1e-10));
    [SpecAugment zeros]; (x - mean) / std with the unbiased standard deviation over all 64 T values."""
    y = np.asarray(y, dtype=np.float64).copy()
    if dither is not None:
        y = y + 1e-5 * np.asarray(dither, dtype=np.float64)
    y = np.concatenate([y[:1], y[1:] - 0.97 * y[:-1]])
    if crop is not None:                               # bug-compatible sub_secquence: the slice ends at target_length (:138-148)
        n = y.shape[0]
        target = int(n * (0.98 + 0.02 * crop[0]))
        loc = int((n - target) * crop[1])
        y = y[loc:target]
    mel = power_spectrum(y, window) @ (filterbank() if fb is None else fb)             # (T, 64)
    db = 10.0 * np.log10(np.maximum(mel, 1e-10)).T     # (64, T); AmplitudeToDB('power'), ref 1, top_db None
    if aug is not None:
        rx, wx, ry, wy = aug
        db[rx:rx + wx, :] = 0.0
        db[:, ry:ry + wy] = 0.0
    if not normalize:
        return db
    n = db.size
    mean = db.sum() / n
    std = math.sqrt(((db - mean) ** 2).sum() / (n - 1))
    return (db - mean) / std
