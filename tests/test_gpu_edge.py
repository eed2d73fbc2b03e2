"""GPU: the edge cases of the decode driver that the ordinary fixtures never reach.

  * beams / greedy rows that FINISH on end-of-text at different depths (transcribe.rs:312-318,
    beam.rs:22-31): finished beams leave the live set, the step batch shrinks, parent slots are
    remapped, the device-chained greedy loop raises per-window done flags;
  * sequences longer than one self-attention tile (> 112 cached positions) and the paged cache tables;
  * the context limit (mod.rs:134-139 panics -> WB_ERR_SHAPE), empty / minimal audio.

Everything is compared token-for-token with the oracle on the same seeded inputs.
"""
import numpy as np
import pytest

import whisper_burn_amd as wb
from oracle import transcribe as otr
from oracle.model import OracleWhisper
from whisper_burn_amd import synth

pytestmark = pytest.mark.gpu


def _special(st: wb.SpecialTokens) -> otr.SpecialTokens:
    return otr.SpecialTokens(st.start_of_transcript, st.language, st.transcribe, st.no_timestamps,
                             st.end_of_text, st.is_special.astype(bool))


@pytest.fixture(scope="module")
def eot_micro():
    """The default synthetic checkpoint: its <|endoftext|> logit ramps up with the position (synth.py), so
    windows / beams finish at different depths."""
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    w = synth.synth_weights(dims, seed=4242)
    return OracleWhisper(w), wb.Whisper.from_tensors(w), wb.SpecialTokens.for_vocab(1031)


eot_micro_strong = eot_micro


@pytest.fixture(scope="module")
def micro():
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    w = synth.synth_weights(dims, seed=4242)
    return OracleWhisper(w), wb.Whisper.from_tensors(w), wb.SpecialTokens.for_vocab(1031)


@pytest.fixture(scope="module")
def micro_no_eot():
    """Without the <|endoftext|> ramp a decode runs to max_depth (long sequences)."""
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    w = synth.synth_weights(dims, seed=4242, eot_beta=0.0)
    return OracleWhisper(w), wb.Whisper.from_tensors(w), wb.SpecialTokens.for_vocab(1031)


@pytest.mark.parametrize("beam_size", [1, 3, 5])
def test_windows_finish_on_end_of_text_at_different_depths(eot_micro, beam_size):
    oracle, eng, st = eot_micro
    audio = synth.synth_audio(16000 * 60, 1236)                  # 60 s -> 6 reference windows
    depth = 10
    ref, ref_win = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, beam_size, depth, return_windows=True)
    ended = [w[-1] == st.end_of_text for w in ref_win]
    assert any(ended) and not all(ended), "fixture no longer mixes finished and unfinished windows"
    got, got_win = wb.waveform_to_tokens(eng, st, audio, 16000, beam_size, depth)
    assert got_win == ref_win
    assert got == ref


@pytest.mark.parametrize("beam_size", [1, 3])
def test_batch_mode_with_finishing_windows(eot_micro_strong, beam_size):
    """> 8 live rows (batch-mode kernels) while rows drop out on end-of-text."""
    oracle, eng, st = eot_micro_strong
    audio = synth.synth_audio(16000 * 150, 777)                  # 13 windows
    depth = 10
    ref, ref_win = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, beam_size, depth, return_windows=True)
    ended = [w[-1] == st.end_of_text for w in ref_win]
    assert sum(ended) >= 4 and not all(ended)
    got, got_win = wb.waveform_to_tokens(eng, st, audio, 16000, beam_size, depth)
    assert got_win == ref_win
    assert got == ref


def test_finishing_windows_are_batch_composition_invariant(eot_micro_strong):
    _, eng, st = eot_micro_strong
    audio = synth.synth_audio(16000 * 150, 777)
    _, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 3, 10)
    p = wb.decode_params(st, beam_size=3, max_depth=10, max_batch_windows=2)
    _, wins2 = wb.waveform_to_tokens(eng, st, audio, 16000, params=p)
    assert wins2 == wins


@pytest.mark.parametrize("beam_size,depth", [(1, 150), (3, 124)])
def test_long_sequences_cross_the_self_attention_tile(micro_no_eot, beam_size, depth):
    """More than 112 cached positions: the second self-attention key tile and the position tables."""
    oracle, eng, st = micro_no_eot
    audio = synth.synth_audio(16000 * 3, 31)
    ref = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, beam_size, depth)
    got, _ = wb.waveform_to_tokens(eng, st, audio, 16000, beam_size, depth)
    assert len(ref) == 4 + depth and len(set(ref[4:])) >= 40
    assert got == ref


def test_decode_without_special_mask_is_a_state_error(micro):
    """The first generated tokens read the special-token mask (transcribe.rs:271-275): a session decode
    before wb_session_set_special_mask must fail like wb_session_step does, on both decode paths."""
    _, eng, st = micro
    audio = synth.synth_audio(16000 * 2, 35)
    for beams in (1, 2):
        sess = wb.Session.begin(eng, audio, [0], [len(audio)], max_beams=beams)
        with pytest.raises(wb.WbError) as e:
            sess.decode(wb.decode_params(st, beam_size=beams, max_depth=4))
        assert e.value.status == -6, e.value.status       # WB_ERR_STATE
        sess.close()
    # a pooled session must not inherit the previous caller's mask either
    sess = wb.Session.begin(eng, audio, [0], [len(audio)], max_beams=1)
    sess.set_special_mask(st.is_special)
    assert len(sess.decode(wb.decode_params(st, beam_size=1, max_depth=4))[0]) >= 5
    sess.close()
    sess = wb.Session.begin(eng, audio, [0], [len(audio)], max_beams=1)
    with pytest.raises(wb.WbError):
        sess.decode(wb.decode_params(st, beam_size=1, max_depth=4))
    sess.close()


def test_large_max_depth_only_fails_when_a_window_outgrows_the_context(micro, micro_no_eot):
    """mod.rs:134-139 fires when a sequence REACHES n_text_ctx, not when max_depth could: with every window
    ending on <|endoftext|> early, max_depth = 1000 succeeds on both greedy paths; without it, it fails."""
    oracle, eng, st = micro
    audio = synth.synth_audio(16000 * 5, 36)
    ref = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, 1, 460)
    assert ref[-1] == st.end_of_text and len(ref) < 60
    got, _ = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 460)
    assert got == ref
    _, eng2, _ = micro_no_eot
    with pytest.raises(wb.WbError) as e:
        wb.waveform_to_tokens(eng2, st, audio, 16000, 1, 460)
    assert e.value.status == -2


@pytest.mark.parametrize("beam_size", [1, 2])
def test_context_limit_is_an_error_not_a_truncation(beam_size):
    """mod.rs:134-139: the decoder panics when the token sequence outgrows n_text_ctx."""
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031, n_text_ctx=16)
    w = synth.synth_weights(dims, seed=7)
    eng, st = wb.Whisper.from_tensors(w), wb.SpecialTokens.for_vocab(1031)
    audio = synth.synth_audio(16000 * 2, 32)
    # depth 13: the decoder sees prefixes of 4 .. 16 tokens (the 17th token is produced, never fed back)
    oracle = OracleWhisper(w)
    ref = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, beam_size, 13)
    got, _ = wb.waveform_to_tokens(eng, st, audio, 16000, beam_size, 13)
    assert got == ref and len(got) == 17
    with pytest.raises(AssertionError):
        otr.waveform_to_tokens(oracle, _special(st), audio, 16000, beam_size, 14)
    with pytest.raises(wb.WbError) as e:
        wb.waveform_to_tokens(eng, st, audio, 16000, beam_size, 14)
    assert e.value.status == -2
    eng.close()


def test_minimal_and_empty_audio(micro):
    oracle, eng, st = micro
    audio = synth.synth_audio(400, 33)                           # exactly n_fft samples -> 1 STFT frame + padding
    ref = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, 1, 6)
    got, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 6)
    assert len(wins) == 1 and got == ref
    for n in (0, 399):                                           # audio.rs:292 panics below n_fft samples
        with pytest.raises(wb.WbError) as e:
            wb.waveform_to_tokens(eng, st, np.zeros(n, np.float32), 16000, 1, 6)
        assert e.value.status == -2, n


def test_max_depth_zero_returns_the_prompt(micro):
    _, eng, st = micro
    audio = synth.synth_audio(16000 * 2, 34)
    for beam in (1, 3):
        got, wins = wb.waveform_to_tokens(eng, st, audio, 16000, beam, 0)
        assert got == [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]


def test_whisper_window_geometry_is_opt_in():
    """wb_model_set_frame_limit: by default a window holds at most n_audio_ctx MEL FRAMES and a longer one is the
    reference's panic (mod.rs:236-241 -> WB_ERR_SHAPE); opted in, it holds 2 n_audio_ctx frames = n_audio_ctx encoder
    positions (Whisper's own 30 s chunk: 12 key chunks per head in the decoder's cross-attention, the kvsplit /
    flash encoder attention over 1500 keys).  Encoder output and beam-3 tokens against the oracle with the same
    switch."""
    import torch
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)           # n_audio_ctx = 1500
    w = synth.synth_weights(dims, seed=23)
    eng, st = wb.Whisper.from_tensors(w), wb.SpecialTokens.for_vocab(1031)
    audio = synth.synth_audio(478559, 52)                        # one 29.9 s window (+ the 3 s tail window its overlap makes)
    mel = np.concatenate([wb.prep_audio(audio[None]), np.zeros((1, 80, 10), np.float32)], 2)
    assert mel.shape == (1, 80, 3000) and eng.max_mel_frames() == 1500
    with pytest.raises(wb.WbError) as e:
        eng.forward_encoder(mel)
    assert e.value.status == -2 and "cannot exceed 1500" in str(e.value)
    eng.set_frame_limit(True)
    assert eng.max_mel_frames() == 3000
    o = OracleWhisper(w, frame_limit_x2=True)
    enc = eng.forward_encoder(mel)
    ref_enc = o.forward_encoder(torch.from_numpy(mel)).numpy()
    assert enc.shape == (1, 1500, 128) and np.abs(enc - ref_enc).max() < 2e-4
    for beam, depth in ((1, 12), (3, 8)):
        got, wins = wb.waveform_to_tokens(eng, st, audio, 16000, beam, depth)
        ref, rw = otr.waveform_to_tokens(o, _special(st), audio, 16000, beam, depth, return_windows=True)
        assert len(wins) == 2 and wins == rw and got == ref, (beam, got, ref)
    eng.set_frame_limit(False)                                                         # and back: the reference's windows
    got, wins = wb.waveform_to_tokens(eng, st, audio[:300000], 16000, 1, 6)
    assert len(wins) == 2 and got == otr.waveform_to_tokens(OracleWhisper(w), _special(st), audio[:300000], 16000, 1, 6)
    eng.close()
