"""GPU: BASELINE.json's configs #2-#5 at their NAMED model and geometry, through the C ABI, against the
oracle's committed rows (tests/golden/oracle_outputs.npz -- the literal decode loop, frozen by
make_golden.py; tests/test_fixture_quality.py proves on the CPU that the oracle reproduces them and that
they are not degenerate).

  #2 tiny.en, 30 s (3 reference windows), greedy, depth 100 -- EXACTLY bench.py's step, hipGraph replay and
     device-chained greedy loop enabled; per window and stitched
     + the reference's live setting on the same audio: beam 5 x depth 100 (transcribe.rs:232-233)
  #3 base.en, 30 s, beam 5, depth 32 (and with windows that end on <|endoftext|>)
  #4 small (V = 51 865, d = 768, 12 layers), 10 minutes = 51 windows: first and last window vs the oracle,
     logits of wb_forward at small's real shape <= 1e-3
  #5 large-v2, one full 14.9 s window (+ its 3 s tail window), greedy
  #2(b) the "perf geometry": tiny.en, ONE 30 s window (T = 2990 + 10, C = 1500), greedy depth 100 -- opt-in on both
     sides (wb_model_set_frame_limit; the reference itself panics on such a window, mod.rs:236-241)

and per-step log-prob parity of the KV-cached session at tiny.en's real shape over 134 positions (the second
112-key self-attention tile, forking / dying beams, two windows of different length).
"""
import os

import numpy as np
import pytest
import torch

import parity_log
import parity_util as pu
import whisper_burn_amd as wb
import workloads
from oracle.model import OracleWhisper

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_outputs.npz")
LOGPROB_TOL = 1e-3
LOGIT_TOL = 1e-3


def rows_of(name):
    g = np.load(GOLD)
    t, n = g[f"{name}_tokens"], g[f"{name}_lens"]
    return [t[i, :n[i]].tolist() for i in range(len(n))]


@pytest.mark.parametrize("name", ["tiny_bench", "tiny_beam5", "base_beam5", "base_beam5_eot", "small_10min",
                                  "large_window", "tiny_whisper30"])
def test_workload_tokens_match_oracle(name):
    wl = workloads.WORKLOADS[name]
    eng = wb.Whisper.from_tensors(wl.weights())
    if wl.frame_limit_x2:          # the opt-in 30 s window (T = 2990, C = 1500: 12 key chunks per head in the decoder)
        eng.set_frame_limit(True)
    st = wb.SpecialTokens.for_vocab(eng.dims["n_vocab"])
    full, wins = wb.waveform_to_tokens(eng, st, wl.audio(), 16000, wl.beam, wl.depth)
    eng.close()
    ref = rows_of(name)
    if wl.windows is None:
        assert len(wins) == len(ref)
        for i, (g, r) in enumerate(zip(wins, ref)):
            assert g == r, (name, "window", i, "first difference at",
                            next((j for j, (a, b) in enumerate(zip(g, r)) if a != b), min(len(g), len(r))))
        assert full == np.load(GOLD)[f"{name}_stitched"].tolist()
    else:
        assert len(wins) == 51
        for r, wi in zip(ref, wl.windows):
            assert wins[wi] == r, (name, "window", wi)
        assert len({tuple(w[4:]) for w in wins}) >= 45          # the 51 windows decode to different text


def test_bench_workload_greedy_chain_and_logprobs_live():
    """bench.py's workload once more, this time checked against a LIVE oracle pass (not the committed rows):
    every generated token is the oracle's argmax given the same prefix (teacher-forced, parity_util), and the
    session's log-prob row at every step is within 1e-3 of the oracle's."""
    wl = workloads.WORKLOADS["tiny_bench"]
    w = wl.weights()
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(51864)
    audio = wl.audio()
    _, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 1, wl.depth)
    mels = pu.window_mels(o, audio, frontend=wb.prep_audio)      # the same log-mel on both sides (see parity_util)
    starts, lens = wb.window_extents(len(audio), 16000, 238559)
    sess = wb.Session.begin(eng, audio, starts, lens, max_beams=1)
    sess.set_special_mask(st.is_special)
    rows_ref = []
    for wi, row in enumerate(wins):
        enc = o.forward_encoder(mels[wi])[0]
        assert np.abs(sess.encoder_output(wi) - enc.numpy()).max() < 1e-4      # measured 8e-6 (max |enc| 4.9)
        lp = pu.teacher_forced_logprobs(o, st, enc, row)
        ok, bad, gap = pu.greedy_chain_report(lp, row, st.end_of_text, wl.depth)
        assert ok, (wi, bad, gap)
        rows_ref.append(lp)
    # replay the three rows through the KV-cached session step by step (rows that ended early keep being fed
    # their last token: their later log-probs are not compared)
    n_pos = max(len(r) for r in wins)
    worst, mag, n_rows = 0.0, 0.0, 0
    for p in range(n_pos - 1):
        toks = [r[min(p, len(r) - 1)] for r in wins]
        par = [-1] * 3 if p == 0 else [0, 1, 2]
        k = 1 if p >= 3 else 0
        sess.step(toks, par, [0, 1, 2], apply_special_mask=(p + 1 <= 5 and p >= 3), k=k)
        if p < 3:
            continue
        for wi, r in enumerate(wins):
            if p <= len(r) - 2:
                got = sess.last_logprobs(wi)
                ref = rows_ref[wi][p - 3]
                fin = np.isfinite(ref)
                assert (np.isfinite(got) == fin).all()
                worst = max(worst, float(np.abs(got[fin] - ref[fin]).max()))
                mag, n_rows = max(mag, float(np.abs(ref[fin]).max())), n_rows + 1
    parity_log.record("workloads::bench_workload_logprobs_live[tiny.en 3 windows depth 100]", worst, LOGPROB_TOL, mag, n_rows=n_rows)
    assert worst < LOGPROB_TOL, worst
    sess.close()
    eng.close()


def test_small_forward_real_shape_logits():
    """wb_forward at small's real shape (V = 51 865, d = 768, 12 + 12 layers): logits <= 1e-3."""
    wl = workloads.WORKLOADS["small_10min"]
    w = wl.weights()
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    assert eng.dims["n_vocab"] == 51865 and eng.dims["n_audio_state"] == 768 and eng.dims["n_text_layer"] == 12
    audio = wl.audio()[:238559]
    mel = np.concatenate([wb.prep_audio(audio[None]), np.zeros((1, 80, 10), np.float32)], 2)
    tokens = np.array([rows_of("small_10min")[0]], dtype=np.int32)
    logits = eng.forward(mel, tokens)
    ref = o.forward(torch.from_numpy(mel), torch.from_numpy(tokens)).numpy()
    assert logits.shape == (1, tokens.shape[1], 51865)
    parity_log.record("workloads::small_forward_real_shape_logits", np.abs(logits - ref).max(), LOGIT_TOL, np.abs(ref).max(),
                      n_rows=tokens.shape[1], quantity="logits")
    assert np.abs(logits - ref).max() < LOGIT_TOL, np.abs(logits - ref).max()
    assert (logits.argmax(-1) == ref.argmax(-1)).all()
    eng.close()


def test_session_logprobs_tiny_real_shape_134_positions():
    """KV-cached beam steps at tiny.en's REAL shape for 134 positions: beams fork and die, two windows of
    different length decode together, every step's full log-prob row of every live beam is compared with
    the stateless oracle (<= 1e-3) and the device top-k with the row's own ordering.  Rows depend on the
    whole history (self-attention over > 112 cached positions = the second key tile, the paged cache
    tables after re-indexing)."""
    wl = workloads.WORKLOADS["tiny_beam5"]
    w = wl.weights()
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(51864)
    audio = wl.audio()
    starts, lens = wb.window_extents(len(audio), 16000, 238559)
    use = [0, 2]                                                  # 14.9 s and 6.2 s
    sess = wb.Session.begin(eng, audio, starts[use], lens[use], max_beams=5)
    sess.set_special_mask(st.is_special)
    mels = pu.window_mels(o, audio, frontend=wb.prep_audio)      # the same log-mel on both sides (see parity_util)
    encs = [o.forward_encoder(mels[i])[0] for i in use]
    prompt = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
    n_steps = 134
    K = 5
    # a beam = (sequence so far, window); slot order = list order
    beams = [([prompt[0]], 0), ([prompt[0]], 1)]
    parents = [-1, -1]
    records = []                                                  # (sequence tuple, window, log-prob row)
    rng = np.random.default_rng(5)
    for step in range(n_steps):
        toks = [b[0][-1] for b in beams]
        wins = [b[1] for b in beams]
        feeding_prompt = step < 3
        use_mask = (not feeding_prompt) and (step + 1 <= 5)
        ids, lps = sess.step(toks, parents, wins, apply_special_mask=use_mask, k=0 if feeding_prompt else K)
        if feeding_prompt:
            beams = [(b[0] + [prompt[step + 1]], b[1]) for b in beams]
            parents = list(range(len(beams)))
            continue
        for slot, (seq, wdx) in enumerate(beams):
            got = sess.last_logprobs(slot)
            order = np.lexsort((np.arange(got.shape[0]), -got.astype(np.float64)))[:K]
            assert ids[slot].tolist() == order.tolist(), (step, slot)
            assert np.allclose(lps[slot], got[ids[slot]], atol=1e-6)
            records.append((tuple(seq), wdx, got))
        # next generation: every beam continues with one of its own top-k (history-dependent choice); every
        # 9th step the first beam of each window forks (while there is room) and every 13th its last beam dies
        nxt, npar = [], []
        count = {0: sum(1 for b in beams if b[1] == 0), 1: sum(1 for b in beams if b[1] == 1)}
        last_of = {wdx: max(i for i, b in enumerate(beams) if b[1] == wdx) for wdx in (0, 1)}
        first_of = {wdx: min(i for i, b in enumerate(beams) if b[1] == wdx) for wdx in (0, 1)}
        for slot, (seq, wdx) in enumerate(beams):
            if step % 13 == 12 and slot == last_of[wdx] and count[wdx] > 1:
                continue
            pick = int(rng.integers(0, K))
            nxt.append((seq + [int(ids[slot][pick])], wdx)); npar.append(slot)
            if step % 9 == 4 and slot == first_of[wdx] and count[wdx] < 5:
                nxt.append((seq + [int(ids[slot][(pick + 1) % K])], wdx)); npar.append(slot)
                count[wdx] += 1
        beams, parents = nxt, npar
    sess.close()
    eng.close()
    assert max(len(r[0]) for r in records) >= 134
    # the oracle: one stateless forward per maximal sequence gives the rows of all its prefixes
    seqs = sorted({(r[0], r[1]) for r in records}, key=lambda x: -len(x[0]))
    rows = {}
    for seq, wdx in seqs:
        if (seq, wdx) in rows:
            continue
        lp = pu.teacher_forced_logprobs(o, st, encs[wdx], list(seq))       # rows for prefixes of length 4 .. len
        for n in range(4, len(seq) + 1):
            rows.setdefault((seq[:n], wdx), lp[n - 4])
    worst, mag, sq = 0.0, 0.0, 0.0
    for seq, wdx, got in records:
        ref = rows[(seq, wdx)]
        fin = np.isfinite(ref)
        assert (np.isfinite(got) == fin).all()
        e = float(np.abs(got[fin] - ref[fin]).max())
        worst, mag, sq = max(worst, e), max(mag, float(np.abs(ref[fin]).max())), sq + e * e
    parity_log.record("workloads::session_logprobs_tiny_real_shape_134_positions", worst, LOGPROB_TOL, mag,
                      rms=(sq / len(records)) ** 0.5, n_rows=len(records))
    assert worst < LOGPROB_TOL, worst
