"""WAV ingest with the reference's sample scaling (src/bin/transcribe/main.rs:31-55, hound reader):
integer PCM s -> s / (2^(bits-1) - 1), float as stored, 16 kHz mono asserted.  The host parser runs
without a GPU; the device conversion kernel is a `gpu` test."""
import os
import struct
import wave

import numpy as np
import pytest

import whisper_burn_amd as wb

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "audio_16k_s16.npz")


def _riff(fmt_chunk: bytes, data: bytes, extra: bytes = b"") -> bytes:
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk + extra
    body += b"data" + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _fmt(tag, channels, rate, bits):
    return struct.pack("<HHIIHH", tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)


def test_golden_s16_through_the_wave_module(tmp_path):
    s16 = np.load(GOLDEN)["pcm"]          # the reference's bundled audio.wav at 16 kHz (tests/golden/make_golden.py)
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(s16.tobytes())
    info = wb.wav_info(path)
    assert info == dict(n_samples=len(s16), sample_rate=16000, channels=1, bits=16, is_float=False)
    got, sr = wb.load_audio_waveform(path)
    assert sr == 16000
    ref = s16.astype(np.float32) / np.float32(32767.0)          # main.rs:45-52: / (2^15 - 1), not / 2^15
    assert got.dtype == np.float32 and np.array_equal(got, ref)


@pytest.mark.parametrize("bits", [8, 16, 24, 32])
def test_integer_widths_follow_hound(tmp_path, bits):
    rng = np.random.default_rng(bits)
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    s = rng.integers(lo, hi + 1, size=1001, dtype=np.int64)
    s[:2] = (lo, hi)
    if bits == 8:
        raw = (s + 128).astype(np.uint8).tobytes()                # 8-bit WAV is unsigned on disk
    elif bits == 16:
        raw = s.astype("<i2").tobytes()
    elif bits == 24:
        raw = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in s)
    else:
        raw = s.astype("<i4").tobytes()
    path = str(tmp_path / f"i{bits}.wav")
    open(path, "wb").write(_riff(_fmt(1, 1, 16000, bits), raw, extra=b"LIST" + struct.pack("<I", 3) + b"abc\0"))
    got, _ = wb.load_audio_waveform(path)
    ref = s.astype(np.float32) / np.float32(float((1 << (bits - 1)) - 1))     # `s as f32 / max_int_val as f32`
    assert np.array_equal(got, ref)


def test_float_and_extensible(tmp_path):
    x = np.linspace(-1.5, 1.5, 777, dtype=np.float32)
    path = str(tmp_path / "f.wav")
    open(path, "wb").write(_riff(_fmt(3, 1, 16000, 32), x.tobytes()))
    got, _ = wb.load_audio_waveform(path)
    assert np.array_equal(got, x)
    # WAVE_FORMAT_EXTENSIBLE wrapping 16-bit PCM
    ext = _fmt(0xFFFE, 1, 16000, 16) + struct.pack("<HHI", 22, 16, 4) + struct.pack("<H", 1) + b"\0" * 14
    s = np.arange(-5, 6, dtype="<i2")
    path2 = str(tmp_path / "e.wav")
    open(path2, "wb").write(_riff(ext, s.tobytes()))
    got2, _ = wb.load_audio_waveform(path2)
    assert np.array_equal(got2, s.astype(np.float32) / np.float32(32767.0))


def test_reference_asserts(tmp_path):
    s = np.zeros(100, "<i2").tobytes()
    p1 = str(tmp_path / "sr.wav"); open(p1, "wb").write(_riff(_fmt(1, 1, 44100, 16), s))
    p2 = str(tmp_path / "ch.wav"); open(p2, "wb").write(_riff(_fmt(1, 2, 16000, 16), s))
    for p, msg in ((p1, "16k"), (p2, "single-channel")):
        with pytest.raises(wb.WbError) as e:
            wb.load_audio_waveform(p)
        assert e.value.status == -2 and msg in str(e.value)     # main.rs:42-43 asserts
    assert wb.wav_info(p1)["sample_rate"] == 44100               # the header query does not judge
    x, sr = wb.load_audio_waveform(p1, any_rate=True)            # the resampler's reader: same scaling, no rate assert
    assert sr == 44100 and x.shape == (100,) and not x.any()
    with pytest.raises(wb.WbError):
        wb.load_audio_waveform(p2, any_rate=True)                # still mono only
    with pytest.raises(wb.WbError) as e:
        wb.load_audio_waveform(str(tmp_path / "missing.wav"))
    assert e.value.status == -3
    p3 = str(tmp_path / "junk.wav"); open(p3, "wb").write(b"not a wave file at all")
    with pytest.raises(wb.WbError) as e:
        wb.load_audio_waveform(p3)
    assert e.value.status == -3


@pytest.mark.gpu
def test_device_conversion_is_bit_identical():
    import torch
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 8, 1001, 480000):
        s = rng.integers(-32768, 32768, size=n, dtype=np.int16)
        if n >= 2:
            s[:2] = (-32768, 32767)
        src = torch.from_numpy(s).cuda()
        dst = torch.full((n + 3,), 9.0, dtype=torch.float32, device="cuda")
        wb.pcm_s16_to_f32_dev(src.data_ptr(), n, dst.data_ptr())
        got = dst.cpu().numpy()
        assert np.array_equal(got[:n], s.astype(np.float32) / np.float32(32767.0))
        assert (got[n:] == 9.0).all()
