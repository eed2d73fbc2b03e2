"""The device resampler (the reference has none -- README.md:69-74 sends non-16 kHz files through sox).
Checker: SciPy's resample_poly, the tool the golden fixture tests/golden/audio_16k_s16.npz was made with.
CPU: the tap design and the output indexing (restated in numpy over the library's own taps).  GPU: the kernel."""
import ctypes as C

import numpy as np
import pytest
from scipy import signal

import whisper_burn_amd as wb
from whisper_burn_amd import _lib

PAIRS = [(22050, 16000), (44100, 16000), (48000, 16000), (8000, 16000), (11025, 16000), (32000, 16000), (16000, 22050)]


def _indexing_restated(x, taps, up, down):
    """y[j] = sum_i x[i] * h[j*down + half - i*up] (resample.hip), in f64 over the library's f32 taps."""
    half = (len(taps) - 1) // 2
    n_out = -(-len(x) * up // down)
    xu = np.zeros(len(x) * up, np.float64)
    xu[::up] = x
    full = np.convolve(xu, taps.astype(np.float64))
    return full[half + np.arange(n_out) * down]


@pytest.mark.parametrize("rate_in,rate_out", PAIRS)
def test_taps_are_scipys_design(rate_in, rate_out):
    taps, up, down = wb.resample_filter(rate_in, rate_out)
    g = np.gcd(rate_in, rate_out)
    assert (up, down) == (rate_out // g, rate_in // g)
    m = max(up, down)
    ref = signal.firwin(20 * m + 1, 1.0 / m, window=("kaiser", 5.0)) * up
    assert taps.shape == ref.shape
    assert np.abs(taps - ref).max() <= 2e-7 * np.abs(ref).max()
    assert abs(taps.sum(dtype=np.float64) - up) <= 1e-4 * up


@pytest.mark.parametrize("rate_in,rate_out", PAIRS[:4])
@pytest.mark.parametrize("n", [1, 440, 441, 3001])
def test_output_indexing_matches_resample_poly(rate_in, rate_out, n):
    rng = np.random.default_rng(n + rate_in)
    x = rng.standard_normal(n)
    taps, up, down = wb.resample_filter(rate_in, rate_out)
    lib = _lib.load()
    assert lib.wb_resample_len(n, rate_in, rate_out) == -(-n * up // down)
    y = _indexing_restated(x, taps, up, down)
    ref = signal.resample_poly(x, up, down)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


def test_argument_errors():
    lib = _lib.load()
    assert lib.wb_resample_len(10, 0, 16000) == -1 and lib.wb_resample_len(-1, 22050, 16000) == -1
    assert lib.wb_resample_len(10, 16001, 16000) == -1               # 16000/16001 does not reduce: ratio above the cap
    assert lib.wb_resample_len(0, 22050, 16000) == 0
    n = C.c_int32(0)
    buf = np.zeros(8, np.float32)
    assert lib.wb_resample_filter(22050, 16000, buf.ctypes.data_as(_lib.c_float_p), 8, C.byref(n), None, None) == -1
    assert n.value == 8821 and b"capacity" in lib.wb_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in,rate_out", PAIRS)
def test_device_resampler_matches_resample_poly(rate_in, rate_out):
    rng = np.random.default_rng(rate_in)
    for n in (1, 441, 22050 * 3 + 17, 1_000_003):
        t = np.arange(n) / rate_in
        x = (0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 3100.0 * t + 1.0)
             + 0.1 * rng.standard_normal(n)).astype(np.float32)
        y = wb.resample(x, rate_in, rate_out)
        g = np.gcd(rate_in, rate_out)
        ref = signal.resample_poly(x.astype(np.float64), rate_out // g, rate_in // g)
        assert y.shape == ref.shape and y.dtype == np.float32
        assert np.abs(y - ref).max() <= 3e-6, (n, np.abs(y - ref).max())      # f32 taps and f32 accumulation of <= 41 terms
        again = wb.resample(x, rate_in, rate_out)
        assert np.array_equal(y, again)                                       # fixed summation order


@pytest.mark.gpu
def test_resampled_s16_fixture_round_trip():
    """The fixture recipe of tests/golden/make_golden.py (s16 / 32767 -> resample 441:320 -> round to s16) on the
    device, on a 22 050 Hz signal: the s16 result is the SciPy one except where the f64 value sits within f32
    rounding of a .5 boundary."""
    rng = np.random.default_rng(5)
    n = 22050 * 8
    s16 = np.clip(np.round(8000 * np.sin(2 * np.pi * 220.0 * np.arange(n) / 22050) + 2000 * rng.standard_normal(n)),
                  -32768, 32767).astype(np.int16)
    xf = s16.astype(np.float64) / 32767.0
    ref = np.clip(np.round(signal.resample_poly(xf, 320, 441) * 32767.0), -32768, 32767).astype(np.int16)
    y = wb.resample(xf.astype(np.float32), 22050, 16000)
    got = np.clip(np.round(y.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    assert got.shape == ref.shape == (128000,)
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 0.02


@pytest.mark.gpu
def test_unity_ratio_is_a_copy():
    x = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    assert np.array_equal(wb.resample(x, 16000, 16000), x)
