"""GPU: the range guards of the 16-bit matrix paths actually TRIP, fail the way the header says, and recover.

The split-precision GEMMs (encoder side: gemm_f16x3.hip; decoder batch mode: decode_batch.hip) carry their f32 activations as
fp16 hi / lo pieces; an activation with |x| >= 65504 has no fp16 representation, the product is inf / NaN, and the kernel
raises a mapped flag word.  Round 5's review: nothing ever tripped these guards in a test.  The checkpoints here have ONE MLP
hidden unit with a bias of 1e5 and an all-zero row in the second matrix: GELU(1e5) = 1e5 times 0 is exactly 0 in f32 (and in
the oracle), so the exact result is the ordinary one -- but on the fp16 pieces 1e5 is inf, inf * 0 = NaN.

  * encoder guard: transparent -- the pass is repeated on the exact-f32 kernel (no error), the model reports "f32" afterwards,
    the output equals the oracle's; through the session API (immediate check) and through wb_waveform_to_tokens (DEFERRED
    check: resolved behind the decode's own synchronisation, the batch decoded again);
  * decoder guard: the call whose rows are invalid fails with WB_ERR_STATE, the model switches to the exact-f32 decoder GEMMs,
    the retry succeeds and equals the oracle -- on the session-step path and on the device-chained greedy path, where the NaN
    rows' top-1 must not become an embedding index (0x7fffffff was gathered as a token id);
  * the guard words are per session: a second session of the same model whose rows are fine is not failed by the first one's trip.
"""
import numpy as np
import pytest
import torch

import whisper_burn_amd as wb
from oracle import transcribe as otr
from oracle.model import OracleWhisper, log_softmax
from whisper_burn_amd import synth
from whisper_burn_amd._lib import WbError

pytestmark = pytest.mark.gpu

WB_ERR_STATE = -6       # include/whisper_hip.h
V = 1031


def _special(st):
    return otr.SpecialTokens(st.start_of_transcript, st.language, st.transcribe, st.no_timestamps, st.end_of_text,
                             st.is_special.astype(bool))


def _weights(where):
    """Micro checkpoint with one MLP hidden unit of `where` ("enc" / "dec") block 1 pushed to 1e5 and its second-matrix
    row zeroed: exact result unchanged, fp16 pieces overflow."""
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=V)
    w = synth.synth_weights(dims, seed=4242, eot_beta=0.0)
    p = ("encoder" if where == "enc" else "decoder") + "/block_1/mlp"
    j = 4 * 128 - 1
    w[p + "/mlp1/weight"][:, j] = 0.0
    w[p + "/mlp1/bias"][j] = 1.0e5
    w[p + "/mlp2/weight"][j, :] = 0.0
    return w


def test_encoder_guard_trips_and_falls_back_transparently():
    w = _weights("enc")
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(V)
    assert eng.encoder_gemm() == "f16x3"
    audio = synth.synth_audio(16000 * 20, 77)                       # 2 reference windows
    ref, ref_win = otr.waveform_to_tokens(o, _special(st), audio, 16000, 1, 12, return_windows=True)
    # deferred check (wb_waveform_to_tokens): no error, right tokens, the model has left the split kernel
    got, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 12)
    assert eng.encoder_gemm() == "f32"
    assert wins == ref_win and got == ref
    eng.close()
    # immediate check (session API): encoder output equals the oracle's although the first pass overflowed
    eng = wb.Whisper.from_tensors(w)
    assert eng.encoder_gemm() == "f16x3"
    starts, lens = wb.window_extents(len(audio), 16000, 238559)
    sess = wb.Session.begin(eng, audio, starts, lens, max_beams=1)
    assert eng.encoder_gemm() == "f32"
    mel = np.concatenate([wb.prep_audio(audio[None, :int(lens[0])]), np.zeros((1, 80, 10), np.float32)], 2)
    enc = o.forward_encoder(torch.from_numpy(mel))[0].numpy()
    got_enc = sess.encoder_output(0)
    assert np.isfinite(got_enc).all() and np.abs(got_enc - enc).max() < 4e-4
    sess.close(); eng.close()


def _windows(n_win, win_len=16000):
    audio = synth.synth_audio(win_len * n_win, 78)
    starts = np.arange(n_win, dtype=np.int64) * win_len
    lens = np.full(n_win, win_len, dtype=np.int64)
    return audio, starts, lens


def test_decoder_guard_on_the_session_step_path_fails_loudly_then_recovers():
    w = _weights("dec")
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(V)
    assert eng.decoder_gemm() == "f16x3"
    n_win = 18                                                       # > 16 live rows at d = 128: batch mode, the skinny GEMM
    audio, starts, lens = _windows(n_win)
    prompt = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]

    def run(sess):
        rows = None
        for p in range(4):
            ids, lps = sess.step([prompt[p]] * n_win, [-1] * n_win if p == 0 else list(range(n_win)), list(range(n_win)),
                                 apply_special_mask=(p == 3), k=1 if p == 3 else 0)
        return ids, sess.last_logprobs(5)

    sess = wb.Session.begin(eng, audio, starts, lens, max_beams=1)
    sess.set_special_mask(st.is_special)
    other = wb.Session.begin(eng, audio[:16000 * 2], starts[:2], lens[:2], max_beams=1)      # 2 rows: fused path, never split
    other.set_special_mask(st.is_special)
    with pytest.raises(WbError) as e:
        run(sess)
    assert e.value.status == WB_ERR_STATE and "fp16" in str(e.value)
    assert eng.decoder_gemm() == "f32"
    sess.close()
    # the bystander session of the same model is not failed by the other one's trip
    for p in range(4):
        other.step([prompt[p]] * 2, [-1, -1] if p == 0 else [0, 1], [0, 1], apply_special_mask=(p == 3), k=1 if p == 3 else 0)
    other.close()
    # retry: a fresh session decodes on the exact-f32 GEMMs and matches the oracle
    sess = wb.Session.begin(eng, audio, starts, lens, max_beams=1)
    sess.set_special_mask(st.is_special)
    ids, row5 = run(sess)
    sess.close(); eng.close()
    mel = np.concatenate([wb.prep_audio(audio[None, 5 * 16000:6 * 16000]), np.zeros((1, 80, 10), np.float32)], 2)
    enc = o.forward_encoder(torch.from_numpy(mel))
    lg = o.forward_decoder(torch.tensor([prompt]), enc)[0, -1]
    lg = lg + torch.tensor(np.where(st.is_special, -np.inf, 0.0), dtype=torch.float32)
    ref = log_softmax(lg, 0).numpy()
    fin = np.isfinite(ref)
    assert (np.isfinite(row5) == fin).all() and np.abs(row5[fin] - ref[fin]).max() < 1e-3
    assert int(ids[5][0]) == int(np.argmax(ref))


def test_decoder_guard_on_the_greedy_chain_path_fails_loudly_then_recovers():
    """The device-chained greedy loop enqueues whole chunks of steps before the host looks: NaN rows must end on
    <|endoftext|> on the device (never gather E[0x7fffffff]) and the call must fail with WB_ERR_STATE; the retry is exact."""
    w = _weights("dec")
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(V)
    n_win = 18
    audio, starts, lens = _windows(n_win)
    params = wb.decode_params(st, beam_size=1, max_depth=10)

    def decode():
        sess = wb.Session.begin(eng, audio, starts, lens, max_beams=1)
        try:
            sess.set_special_mask(st.is_special)
            return sess.decode(params)
        finally:
            sess.close()

    with pytest.raises(WbError) as e:
        decode()
    assert e.value.status == WB_ERR_STATE
    assert eng.decoder_gemm() == "f32"
    rows = decode()
    eng.close()
    ost = _special(st)
    for wi in (0, 7, 17):
        mel = torch.from_numpy(wb.prep_audio(audio[None, wi * 16000:(wi + 1) * 16000]))
        assert rows[wi] == otr.mels_to_tokens(o, ost, mel, 10, 1, 10), wi
