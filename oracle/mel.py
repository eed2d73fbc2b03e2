"""Oracle (test infrastructure): CPU restatement of the log-mel frontend.

Follows /root/reference/src/audio.rs and src/helper.rs op for op in fp32 on
PyTorch-CPU (burn-tch is a thin wrapper over the same libtorch operators), and
provides a NumPy f64 "exact" twin used to separate "differs from the
reference's rounding" from "wrong".

Scalar ops with an f64 argument (mul_scalar/add_scalar/sub_scalar) convert the
scalar to the element type (f32) before the op, as burn-tch does.
"""
from __future__ import annotations

import math

import numpy as np
import torch

N_FFT = 400          # audio.rs:5
HOP_LENGTH = 160     # audio.rs:6
N_MELS = 80          # audio.rs:7
WINDOW_LENGTH = N_FFT  # audio.rs:8


def max_waveform_samples(n_frame_max: int) -> int:
    """audio.rs:12-17."""
    n_samples_max = HOP_LENGTH * (n_frame_max + 1) + (N_FFT % 2)
    return n_samples_max - 1


def _f32(x: float) -> float:
    """An f64 scalar as burn-tch hands it to an f32 tensor op."""
    return float(np.float32(x))


# ---- helper.rs -----------------------------------------------------------

def tensor_max_scalar(x: torch.Tensor, m: float) -> torch.Tensor:
    """helper.rs:8-10: relu(x - m) + m (NOT a true max: rounds differently)."""
    m = _f32(m)
    return torch.relu(x - m) + m


def tensor_max(x: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """helper.rs:16-18."""
    return torch.relu(x - m) + m


def tensor_min(x: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """helper.rs:20-22."""
    return -tensor_max(-x, -m)


def tensor_log10(x: torch.Tensor) -> torch.Tensor:
    """helper.rs:24-27: ln(x) / ln(10) with ln(10) rounded to f32."""
    return torch.log(x) / _f32(math.log(10.0))


def reverse(x: torch.Tensor, dim: int) -> torch.Tensor:
    """helper.rs:38-48."""
    n = x.shape[dim]
    idx = -torch.arange(n) + (n - 1)
    return x.index_select(dim, idx)


# ---- audio.rs --------------------------------------------------------------

def hann_window(window_length: int = WINDOW_LENGTH) -> torch.Tensor:
    """audio.rs:272-278: sin(n * pi/len)^2 in f32."""
    n = torch.arange(window_length).float()
    return torch.sin(n * _f32(math.pi / window_length)).pow(2.0)


def hz_to_mel(freq: float, htk: bool = False) -> float:
    """audio.rs:198-230 (host f64)."""
    if htk:
        return 2595.0 * math.log10(1.0 + freq / 700.0)
    f_min = 0.0
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = math.log(6.4) / 27.0
    if freq >= min_log_hz:
        return min_log_mel + math.log(freq / min_log_hz) / logstep
    return (freq - f_min) / f_sp


def mel_to_hz_tensor(mel: torch.Tensor) -> torch.Tensor:
    """audio.rs:232-266 (htk=false branch), f32 tensor ops."""
    f_min = 0.0
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = math.log(6.4) / 27.0
    log_t = (mel >= _f32(min_log_mel)).float()
    freq = log_t * (torch.exp((mel - _f32(min_log_mel)) * _f32(logstep)) * _f32(min_log_hz)) \
        + (-log_t + 1.0) * (mel * _f32(f_sp) + _f32(f_min))
    return freq


def mel_frequencies(n_mels: int, fmin: float, fmax: float) -> torch.Tensor:
    """audio.rs:178-196."""
    min_mel = hz_to_mel(fmin)
    max_mel = hz_to_mel(fmax)
    mels = torch.arange(n_mels).float() * _f32((max_mel - min_mel) / (n_mels - 1)) + _f32(min_mel)
    return mel_to_hz_tensor(mels)


def fft_frequencies(sample_rate: float, n_fft: int) -> torch.Tensor:
    """audio.rs:149-157."""
    return torch.arange(n_fft // 2 + 1).float() * _f32(sample_rate / n_fft)


def get_mel_filters(sample_rate: float, n_fft: int = N_FFT, n_mels: int = N_MELS) -> torch.Tensor:
    """audio.rs:67-143 (htk=false): Slaney mel filterbank [n_mels, n_fft/2+1], f32."""
    fmin = 0.0
    fmax = sample_rate * 0.5
    fftfreqs = fft_frequencies(sample_rate, n_fft)
    mel_f_size = n_mels + 2
    mel_f = mel_frequencies(mel_f_size, fmin, fmax)
    fdiff = mel_f[1:mel_f_size] - mel_f[0:mel_f_size - 1]
    ramps = mel_f[:, None].repeat(1, fftfreqs.shape[0]) - fftfreqs[None, :]
    lower = -ramps[0:n_mels] / fdiff[0:n_mels][:, None]
    upper = ramps[2:2 + n_mels] / fdiff[1:1 + n_mels][:, None]
    weights = torch.relu(tensor_min(lower, upper))
    enorm = (mel_f[2:n_mels + 2] - mel_f[0:n_mels]).pow(-1.0) * 2.0
    weights = weights * enorm[:, None]
    return weights


def stfft(x: torch.Tensor, n_fft: int = N_FFT, hop_length: int = HOP_LENGTH,
          window: torch.Tensor | None = None):
    """audio.rs:284-367: reflect-pad, frame, dense DFT by two f32 matmuls.

    x: [B, N] f32 -> (real, imag) each [B, n_fft/2+1, N/hop+1].
    """
    if window is None:
        window = hann_window(n_fft)
    n_batch, n = x.shape
    assert n >= n_fft                                   # audio.rs:292
    pad = n_fft // 2
    left_pad = reverse(x[:, 1:pad + 1], 1)             # audio.rs:298
    right_pad = reverse(x[:, n - pad - 1:n - 1], 1)    # audio.rs:299-305
    xp = torch.cat([left_pad, x, right_pad], 1)
    input_size = xp.shape[1]
    n_frame = (input_size - n_fft) // hop_length + 1
    n_freq = n_fft // 2 + 1
    # audio.rs:331-346 builds input_windows[b, n, f] = xp[b, f*hop + n] through a
    # reshape/transposes/cat of shifted slices; as_strided yields the same values.
    input_windows = xp.as_strided((n_batch, n_fft, n_frame),
                                  (xp.stride(0), 1, hop_length)).contiguous()
    coe = math.pi * 2.0 / n_fft
    b = (torch.arange(n_freq).float() * _f32(coe))[:, None].repeat(1, n_fft) \
        * torch.arange(n_fft).float()[None, :]          # audio.rs:349-356
    real = (torch.cos(b) * window[None, :])[None].matmul(input_windows)       # :359-361
    imag = (torch.sin(b) * (-window)[None, :])[None].matmul(input_windows)    # :362-364
    return real, imag


def prep_audio(waveform: torch.Tensor, sample_rate: float = 16000.0) -> torch.Tensor:
    """audio.rs:34-56: [B, N] f32 -> [B, 80, N // 160] f32."""
    waveform = waveform.float()
    window = hann_window(WINDOW_LENGTH)
    re, im = stfft(waveform, N_FFT, HOP_LENGTH, window)
    magnitudes = re.pow(2.0) + im.pow(2.0)
    magnitudes = magnitudes[:, :, :magnitudes.shape[2] - 1]              # :41-42
    mel_spec = get_mel_filters(sample_rate, N_FFT, N_MELS)[None].matmul(magnitudes)
    log_spec = tensor_log10(tensor_max_scalar(mel_spec, 1.0e-10))         # :48
    mx = float(log_spec.max())                                            # :50 (f32 -> f64)
    log_spec = tensor_max_scalar(log_spec, mx - 8.0)                      # :52
    log_spec = (log_spec + 4.0) / 4.0                                     # :53
    return log_spec


# ---- exact twin (f64) ------------------------------------------------------

def mel_filters_f64(sample_rate: float = 16000.0, n_fft: int = N_FFT, n_mels: int = N_MELS) -> np.ndarray:
    """The same Slaney filterbank evaluated in f64 (librosa.filters.mel algorithm)."""
    fftfreqs = np.arange(n_fft // 2 + 1, dtype=np.float64) * (sample_rate / n_fft)
    min_mel, max_mel = hz_to_mel(0.0), hz_to_mel(sample_rate * 0.5)
    mels = np.arange(n_mels + 2, dtype=np.float64) * ((max_mel - min_mel) / (n_mels + 1)) + min_mel
    f_sp = 200.0 / 3.0
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0
    mel_f = np.where(mels >= min_log_mel, 1000.0 * np.exp(logstep * (mels - min_log_mel)), f_sp * mels)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:n_mels] / fdiff[:n_mels, None]
    upper = ramps[2:n_mels + 2] / fdiff[1:n_mels + 1, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


def prep_audio_f64(waveform: np.ndarray, sample_rate: float = 16000.0) -> np.ndarray:
    """Mathematically exact log-mel ([N] -> [80, N // 160]) in f64 with a real FFT."""
    x = np.asarray(waveform, dtype=np.float64)
    n = x.shape[0]
    assert n >= N_FFT
    xp = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    n_frame = (xp.shape[0] - N_FFT) // HOP_LENGTH + 1
    frames = np.lib.stride_tricks.as_strided(
        xp, (n_frame, N_FFT), (xp.strides[0] * HOP_LENGTH, xp.strides[0]))
    win = np.sin(np.arange(N_FFT) * (math.pi / N_FFT)) ** 2
    spec = np.fft.rfft(frames * win[None, :], axis=1)           # [n_frame, 201]
    power = (spec.real ** 2 + spec.imag ** 2)[:n_frame - 1].T    # [201, T]
    mel = mel_filters_f64(sample_rate) @ power
    log_spec = np.log10(np.maximum(mel, 1e-10))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0
