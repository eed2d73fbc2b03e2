"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on seeded inputs.

Tolerances (BASELINE.json north_star): logits within 1e-3 (f32 path), identical greedy
token ids; mel within 2e-4 of the reference's own f32 recipe (whose dense-DFT angles carry
~7.5e-5 rad of f32 error) and within 5e-5 of the exact f64 result.
"""
import numpy as np
import pytest

import parity_log
import torch

import whisper_burn_amd as wb
from oracle import mel as omel
from oracle.model import OracleWhisper
from whisper_burn_amd import synth

pytestmark = pytest.mark.gpu

MEL_TOL_VS_ORACLE = 2e-4
MEL_TOL_VS_EXACT = 5e-5
ENC_TOL = 2e-4
LOGIT_TOL = 1e-3


@pytest.fixture(scope="module")
def micro():
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    w = synth.synth_weights(dims, seed=4242)
    return w, OracleWhisper(w), wb.Whisper.from_tensors(w)


@pytest.fixture(scope="module")
def tiny():
    w = synth.synth_preset("tiny.en")
    return w, OracleWhisper(w), wb.Whisper.from_tensors(w)


@pytest.mark.parametrize("n,seed", [(238559, 1236), (98882, 1237), (400, 1), (401, 2), (1000, 3), (16000, 4)])
def test_prep_audio_matches_oracle(n, seed):
    x = synth.synth_audio(n, seed)
    got = wb.prep_audio(x[None])[0]
    ref = omel.prep_audio(torch.from_numpy(x)[None])[0].numpy()
    exact = omel.prep_audio_f64(x)
    assert got.shape == ref.shape == (80, n // 160)
    if got.size == 0:
        return
    assert np.abs(got - exact).max() < MEL_TOL_VS_EXACT, np.abs(got - exact).max()
    assert np.abs(got - ref).max() < MEL_TOL_VS_ORACLE, np.abs(got - ref).max()


def test_batched_device_frontend_matches_oracle():
    """wb_waveform_to_mels_dev: reference windowing (transcribe.rs:114-138) + clip / zero-pad (:171-177),
    device pointers in and out, equal to prep_audio on every window."""
    n = 238559 + 2 * 190559 + 5000                     # 4 reference windows, the last one short
    x = synth.synth_audio(n, 77)
    starts, lens = wb.window_extents(n, 16000, 238559)
    assert len(starts) == 4
    Ts = 1500
    pcm = torch.from_numpy(x).cuda()
    out = torch.full((4, 80, Ts), 7.0, dtype=torch.float32, device="cuda")
    frames, ms = wb.waveform_to_mels_dev(pcm.data_ptr(), n, starts, lens, out.data_ptr(), 80 * Ts, Ts)
    got = out.cpu().numpy()
    assert ms > 0
    for w in range(4):
        exact = omel.prep_audio_f64(x[starts[w]:starts[w] + lens[w]])
        keep = min(exact.shape[1], 1490)
        assert frames[w] == keep + 10
        assert np.abs(got[w, :, :keep] - exact[:, :keep]).max() < MEL_TOL_VS_EXACT
        assert (got[w, :, keep:keep + 10] == 0).all()              # padding frames are zeros in log-mel space
    with pytest.raises(wb.WbError) as e:                            # row stride too small for 1500 frames
        wb.waveform_to_mels_dev(pcm.data_ptr(), n, starts, lens, out.data_ptr(), 80 * Ts, 1496)
    assert e.value.status == -1


def test_prep_audio_tone_and_digital_silence():
    # bins on the 1e-10 floor: the reference's f32 recipe itself is only ~1e-3 accurate there
    t = np.arange(48000) / 16000.0
    x = (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    x[16000:24000] = 0.0
    got = wb.prep_audio(x[None])[0]
    exact = omel.prep_audio_f64(x)
    ref = omel.prep_audio(torch.from_numpy(x)[None])[0].numpy()
    assert np.abs(got - exact).max() <= max(2e-3, np.abs(ref - exact).max())
    assert np.mean(np.abs(got - exact) > 1e-4) < 0.02


def test_prep_audio_rejects_short_window():
    with pytest.raises(wb.WbError) as e:
        wb.prep_audio(np.zeros((1, 399), np.float32))
    assert e.value.status == -2       # audio.rs:292 assert


@pytest.mark.parametrize("B,T", [(1, 1500), (2, 628), (1, 7), (3, 33)])
def test_forward_encoder_micro(micro, B, T):
    _, oracle, eng = micro
    mel = np.random.default_rng(T).standard_normal((B, 80, T)).astype(np.float32) * 0.5
    got = eng.forward_encoder(mel)
    ref = oracle.forward_encoder(torch.from_numpy(mel)).numpy()
    assert got.shape == ref.shape == (B, (T - 1) // 2 + 1, 128)
    assert np.abs(got - ref).max() < ENC_TOL, np.abs(got - ref).max()


def test_forward_encoder_rejects_long_input(micro):
    _, _, eng = micro
    with pytest.raises(wb.WbError) as e:
        eng.forward_encoder(np.zeros((1, 80, 1501), np.float32))
    assert e.value.status == -2       # mod.rs:236-241


def test_forward_decoder_micro(micro):
    _, oracle, eng = micro
    rng = np.random.default_rng(5)
    enc = rng.standard_normal((3, 150, 128)).astype(np.float32)
    tokens = rng.integers(0, 1031, (3, 17)).astype(np.int32)
    got = eng.forward_decoder(tokens, enc)
    ref = oracle.forward_decoder(torch.from_numpy(tokens), torch.from_numpy(enc)).numpy()
    assert got.shape == ref.shape == (3, 17, 1031)
    parity_log.record("parity::forward_decoder_micro", np.abs(got - ref).max(), LOGIT_TOL, np.abs(ref).max(), n_rows=51, quantity="logits")
    assert np.abs(got - ref).max() < LOGIT_TOL, np.abs(got - ref).max()
    assert (got.argmax(-1) == ref.argmax(-1)).all()


def test_forward_decoder_rejects_long_prefix(micro):
    _, _, eng = micro
    with pytest.raises(wb.WbError) as e:
        eng.forward_decoder(np.zeros((1, 449), np.int32), np.zeros((1, 10, 128), np.float32))
    assert e.value.status == -2       # mod.rs:134-139


def test_layernorm_variant_switch(micro):
    w, _, eng = micro
    o_in = OracleWhisper(w, ln_eps_inside_sqrt=True)
    mel = np.random.default_rng(9).standard_normal((1, 80, 100)).astype(np.float32) * 0.5
    try:
        eng.set_layernorm_variant(True)
        got = eng.forward_encoder(mel)
    finally:
        eng.set_layernorm_variant(False)
    ref = o_in.forward_encoder(torch.from_numpy(mel)).numpy()
    assert np.abs(got - ref).max() < ENC_TOL


def test_tiny_en_forward_real_shape(tiny):
    _, oracle, eng = tiny
    x = synth.synth_audio(238559, 1236)
    mel = wb.prep_audio(x[None])                                      # [1, 80, 1490]
    mel = np.concatenate([mel, np.zeros((1, 80, 10), np.float32)], 2)  # transcribe.rs:171-177
    tokens = np.array([[50257, 50258, 50358, 50362, 1000, 2000, 3000]], dtype=np.int32)
    enc = eng.forward_encoder(mel)
    ref_enc = oracle.forward_encoder(torch.from_numpy(mel)).numpy()
    assert enc.shape == (1, 750, 384)
    assert np.abs(enc - ref_enc).max() < ENC_TOL * 2, np.abs(enc - ref_enc).max()
    logits = eng.forward(mel, tokens)
    ref = oracle.forward(torch.from_numpy(mel), torch.from_numpy(tokens)).numpy()
    assert logits.shape == (1, 7, 51864)
    parity_log.record("parity::tiny_en_forward_real_shape encoder", np.abs(enc - ref_enc).max(), ENC_TOL * 2, np.abs(ref_enc).max(), quantity="encoder output")
    parity_log.record("parity::tiny_en_forward_real_shape logits", np.abs(logits - ref).max(), LOGIT_TOL, np.abs(ref).max(), n_rows=7, quantity="logits")
    assert np.abs(logits - ref).max() < LOGIT_TOL, np.abs(logits - ref).max()
    assert (logits.argmax(-1) == ref.argmax(-1)).all()
