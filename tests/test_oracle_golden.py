"""CPU: the oracle reproduces the committed golden vectors (tests/golden/, made by make_golden.py
from the reference's audio.wav fixture), so oracle drift cannot silently move the parity target."""
import os

import numpy as np
import torch

from oracle import mel as omel
from oracle import transcribe as otr
from oracle.model import OracleWhisper, log_softmax
from whisper_burn_amd import synth
from whisper_burn_amd.tokens import SpecialTokens

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden():
    g = dict(np.load(os.path.join(GOLD, "oracle_outputs.npz")))
    pcm = np.load(os.path.join(GOLD, "audio_16k_s16.npz"))["pcm"]
    audio = pcm.astype(np.float32) / np.float32(32767.0)      # src/bin/transcribe/main.rs:44-51
    return g, audio


def _ost(st):
    return otr.SpecialTokens(st.start_of_transcript, st.language, st.transcribe, st.no_timestamps,
                             st.end_of_text, st.is_special.astype(bool))


def test_fixture_is_the_resampled_reference_wav():
    _, audio = golden()
    assert audio.shape == (122276,) and audio.dtype == np.float32          # 7.64 s at 16 kHz
    assert 0.05 < np.abs(audio).max() <= 1.0


def test_oracle_mel_matches_golden():
    g, audio = golden()
    mel = omel.prep_audio(torch.from_numpy(audio)[None])[0].numpy()
    assert list(mel.shape) == g["mel_shape"].tolist() == [80, 764]
    assert np.abs(mel[:, :160] - g["mel_head"]).max() < 2e-5
    assert np.abs(mel[::4, ::9] - g["mel_strided"]).max() < 2e-5
    # the reference's f32 dense-DFT recipe vs the exact result: ~1e-3 on the few bins near the
    # max-8 clamp floor (this clip has near-silent high bands), <= 1e-4 elsewhere
    d = np.abs(omel.prep_audio_f64(audio)[:, :160] - g["mel_head"])
    assert d.max() < 2e-3 and np.mean(d > 1e-4) < 0.05


def test_oracle_micro_model_matches_golden():
    g, audio = golden()
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    o = OracleWhisper(synth.synth_weights(dims, seed=4242))
    st = SpecialTokens.for_vocab(1031)
    mel = omel.prep_audio(torch.from_numpy(audio)[None])
    enc = o.forward_encoder(torch.cat([mel, torch.zeros(1, 80, 10)], 2))
    assert list(enc.shape) == g["micro_enc_shape"].tolist()
    assert np.abs(enc[0, ::6, ::3].numpy() - g["micro_enc_strided"]).max() < 1e-4
    lp = log_softmax(o.forward_decoder(torch.from_numpy(g["micro_prefix"]), enc)[0], 1)
    v, ix = torch.topk(lp, 8, dim=1)
    assert np.array_equal(ix.numpy(), g["micro_top_id"])
    assert np.abs(v.numpy() - g["micro_top_lp"]).max() < 1e-3
    assert otr.waveform_to_tokens(o, _ost(st), audio, 16000, 1, 24) == g["micro_greedy"].tolist()
    assert otr.waveform_to_tokens(o, _ost(st), audio, 16000, 5, 24) == g["micro_beam5"].tolist()


def test_oracle_tiny_en_tokens_match_golden():
    """Config #1: the reference's bundled audio.wav, tiny.en, greedy to max_depth 100 -- the literal loop."""
    g, audio = golden()
    o = OracleWhisper(synth.synth_preset("tiny.en"))
    st = SpecialTokens.for_vocab(51864)
    assert otr.waveform_to_tokens(o, _ost(st), audio, 16000, 1, 100) == g["tiny_en_wav_greedy"].tolist()
    assert g["tiny_en_wav_greedy"][:4].tolist() == [50257, 50258, 50358, 50362]
    on = OracleWhisper(synth.synth_preset("tiny.en", eot_beta=0.0))
    assert otr.waveform_to_tokens(on, _ost(st), audio, 16000, 1, 100) == g["tiny_en_wav_greedy_long"].tolist()
    assert len(set(g["tiny_en_wav_greedy_long"][4:].tolist())) >= 50
