"""GPU parity of the KV-cached session / decode driver against the oracle's stateless,
as-written restatement (full-prefix decoder re-run per step, transcribe.rs:253-307)."""
import numpy as np
import pytest

import parity_log
import torch

import whisper_burn_amd as wb
from oracle import transcribe as otr
from oracle.model import OracleWhisper, log_softmax
from whisper_burn_amd import synth

pytestmark = pytest.mark.gpu

LOGPROB_TOL = 1e-3


def _special(st: wb.SpecialTokens) -> otr.SpecialTokens:
    return otr.SpecialTokens(st.start_of_transcript, st.language, st.transcribe, st.no_timestamps,
                             st.end_of_text, st.is_special.astype(bool))


@pytest.fixture(scope="module")
def micro():
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    w = synth.synth_weights(dims, seed=4242)
    return OracleWhisper(w), wb.Whisper.from_tensors(w), wb.SpecialTokens.for_vocab(1031)


@pytest.fixture(scope="module")
def tiny():
    w = synth.synth_preset("tiny.en")
    return OracleWhisper(w), wb.Whisper.from_tensors(w), wb.SpecialTokens.for_vocab(51864)


def _topk_ref(lp: np.ndarray, k: int):
    order = np.lexsort((np.arange(lp.shape[0]), -lp.astype(np.float64)))   # value desc, id asc
    return order[:k]


def test_session_steps_match_stateless_decoder(micro):
    oracle, eng, st = micro
    rng = np.random.default_rng(11)
    mels = [rng.standard_normal((80, 300)).astype(np.float32) * 0.5,
            rng.standard_normal((80, 77)).astype(np.float32) * 0.5]
    padding = 10
    sess = wb.Session.begin_mel(eng, mels, max_beams=3, padding=padding)
    sess.set_special_mask(st.is_special)
    encs = []
    for w, mel in enumerate(mels):
        m = np.concatenate([mel, np.zeros((80, padding), np.float32)], 1)[None]
        ref = oracle.forward_encoder(torch.from_numpy(m))[0]
        got = sess.encoder_output(w)
        assert got.shape == tuple(ref.shape)
        assert np.abs(got - ref.numpy()).max() < 2e-4
        encs.append(ref)
    # scripted beams: (tokens, window); every step extends / forks beams of the previous step
    beams = [([5], 0), ([7], 1)]
    parents = [-1, -1]
    worst, mag, n_rows = 0.0, 0.0, 0
    for step in range(9):
        toks = [b[0][-1] for b in beams]
        wins = [b[1] for b in beams]
        use_mask = step in (2, 3)
        ids, lps = sess.step(toks, parents, wins, apply_special_mask=use_mask, k=4)
        for slot, (seq, w) in enumerate(beams):
            logits = oracle.forward_decoder(torch.tensor([seq]), encs[w][None])[0, -1]
            if use_mask:
                logits = logits + torch.tensor(np.where(st.is_special, -np.inf, 0.0), dtype=torch.float32)
            ref = log_softmax(logits, 0).numpy()
            got = sess.last_logprobs(slot)
            fin = np.isfinite(ref)
            assert (np.isfinite(got) == fin).all()
            worst, mag, n_rows = max(worst, float(np.abs(got[fin] - ref[fin]).max())), max(mag, float(np.abs(ref[fin]).max())), n_rows + 1
            assert np.abs(got[fin] - ref[fin]).max() < LOGPROB_TOL, (step, slot)
            assert ids[slot].tolist() == _topk_ref(got, 4).tolist()
            assert np.allclose(lps[slot], got[ids[slot]], atol=1e-6)
            assert ids[slot][0] == _topk_ref(ref, 1)[0]
        # next generation: fork slot 0 twice while room, extend the rest, drop one beam now and then
        new_beams, new_parents = [], []
        for slot, (seq, w) in enumerate(beams):
            kids = 2 if (slot == 0 and sum(1 for b in new_beams if b[1] == w) + 2 <= 3 and step % 2 == 0) else 1
            if step == 5 and slot == len(beams) - 1 and len(beams) > 2:
                continue
            for j in range(kids):
                new_beams.append((seq + [int(ids[slot][j])], w))
                new_parents.append(slot)
        # keep at most 3 beams per window
        keep = []
        cnt = {0: 0, 1: 0}
        for i, b in enumerate(new_beams):
            if cnt[b[1]] < 3:
                cnt[b[1]] += 1
                keep.append(i)
        beams = [new_beams[i] for i in keep]
        parents = [new_parents[i] for i in keep]
    sess.close()
    parity_log.record("session::scripted_beams_micro_9_steps", worst, LOGPROB_TOL, mag, n_rows=n_rows)


@pytest.mark.parametrize("beam_size", [1, 5])
def test_waveform_to_tokens_matches_oracle_micro(micro, beam_size):
    oracle, eng, st = micro
    audio = synth.synth_audio(480000, 1236)          # 30 s -> 3 reference windows (T = 1490, 1490, 618)
    depth = 20
    ref, ref_win = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, beam_size, depth, return_windows=True)
    got, got_win = wb.waveform_to_tokens(eng, st, audio, 16000, beam_size, depth)
    assert len(got_win) == 3
    assert got_win == ref_win
    assert got == ref


def test_window_sharding_is_exact(micro):
    _, eng, st = micro
    audio = synth.synth_audio(16000 * 40, 99)        # 4 windows
    full, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 8)
    assert len(wins) == 4
    parts = []
    for lo, hi in ((0, 2), (2, 3), (3, 4)):
        _, w = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 8, win_begin=lo, win_end=hi)
        parts += w
    assert parts == wins
    rows = np.zeros((4, 16), np.int32)
    lens = np.array([len(w) for w in wins], np.int32)
    for i, w in enumerate(wins):
        rows[i, :len(w)] = w
    assert wb.stitch_windows(rows, lens) == full


def test_short_last_window_is_rejected(micro):
    _, eng, st = micro
    # 190559 + 100 samples -> second window has 100 samples < n_fft: the reference panics (audio.rs:292)
    audio = synth.synth_audio(190559 + 100, 5)
    with pytest.raises(wb.WbError) as e:
        wb.waveform_to_tokens(eng, st, audio, 16000, 1, 4)
    assert e.value.status == -2


def test_tiny_en_greedy_matches_oracle(tiny):
    oracle, eng, st = tiny
    audio = synth.synth_audio(16000 * 6, 1237)       # one 6 s window, C = 305
    ref = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, 1, 16)
    got, _ = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 16)
    assert got == ref
    assert got[:4] == [50257, 50258, 50358, 50362] and len(got) == 20


def test_tiny_en_beam5_matches_oracle(tiny):
    oracle, eng, st = tiny
    audio = synth.synth_audio(16000 * 4, 1238)
    ref = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, 5, 8)
    got, _ = wb.waveform_to_tokens(eng, st, audio, 16000, 5, 8)
    assert got == ref


@pytest.mark.parametrize("beam_size", [1, 3])
def test_many_windows_batch_mode_matches_oracle(micro, beam_size):
    """> 8 live beams: the batch-mode decode path (split-K MFMA GEMMs, stand-alone LayerNorm, row top-k)
    and, for beam 1, the device-chained greedy loop over more than 8 windows."""
    oracle, eng, st = micro
    audio = synth.synth_audio(16000 * 150, 4321)      # 150 s -> 13 reference windows
    depth = 10
    ref, ref_win = otr.waveform_to_tokens(oracle, _special(st), audio, 16000, beam_size, depth, return_windows=True)
    got, got_win = wb.waveform_to_tokens(eng, st, audio, 16000, beam_size, depth)
    assert len(got_win) == 13
    assert got_win == ref_win
    assert got == ref
