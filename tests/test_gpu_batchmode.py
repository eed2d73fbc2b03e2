"""GPU: the BATCH-MODE decode path (more than 8 live rows: resolve-LN + split-K MFMA GEMMs + paged self-attention +
streaming / chunked cross-attention + top-k rows) against the oracle at the REAL shapes of BASELINE configs #4 / #5,
at the reference's live depth (max_depth 100, transcribe.rs:232-233), through the C ABI.

The oracle side is teacher-forced (tests/parity_util.py): the reference re-runs the stateless decoder over the whole
prefix at every step (transcribe.rs:253-307, mod.rs:345-350), and the decoder is causal, so ONE stateless oracle
forward over a finished row yields the log-prob row of every step -- a row is the oracle's own greedy output iff every
generated token is the (lowest-id) argmax of its row and the row stops where beam.rs:22-31 with k = 1 stops.

  small,    10 min = 51 windows decoded in ONE batch, greedy, depth 100: every window's row is the oracle's chain
  large-v2, 118.75 s = 10 windows in one batch (> 8 rows: the split-K GEMM / stream / self-attention / top-k-rows path
            at d = 1280, 32 layers), greedy, depth 100: every row teacher-forced; plus a 5-window x 2-beam session
            (dec_cross_attn_kernel<2> + combine at d = 1280) whose per-step log-prob rows are <= 1e-3 of the oracle
  small,    batch-mode sessions (9 windows x 1 beam = streaming cross-attention; 5 windows x 2 beams = chunked +
            combine) for 122 positions -- past position 112, the second self-attention tile of dec_self_attn_kernel

Log-prob tolerance: the north star's 1e-3, asserted OUTRIGHT against the f32 oracle at every size (LOGPROB_TOL below; no
ratio, no budget factor).  Rounds 3 / 4 could not do that at large-v2: the synthetic checkpoint amplified a 1e-7 perturbation
190-fold through its 32 cross-attention layers, two correct f32 evaluations differed by up to 3e-2, and the rows were gated on
a ratio of two noisy 41-row samples against an f64 twin -- which failed on the driver's box in round 4.  Round 5 fixed the
FIXTURE instead (synth.synth_weights, "Depth normalisation"): the f32 oracle now sits 1e-5 - 8e-5 from the f64 evaluation of
the same operators at `small` and large-v2 (/tmp study recorded in LABLOG R5.1), so |hip - oracle_f32| <= 1e-3 is a plain
assertion with an order of magnitude of headroom.  Token parity of the batch-mode path is pinned separately and exactly by the
two depth-100 greedy tests.
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
from whisper_burn_amd import synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_outputs.npz")
LOGPROB_TOL = 1e-3      # north_star: logits within 1e-3 (fp32) -- asserted outright on every compared row
WLEN = 238559           # max_waveform_samples(1500 - 10), transcribe.rs:32-34


def _check_rows_are_oracle_greedy(o, st, audio, wins, depth, sample_rate=16000, only=None):
    """Every per-window row (`only`: these window indices) is the oracle's greedy chain given the same log-mel
    (parity_util.window_mels feeds both sides the HIP log-mel; the frontend has its own tests).
    Returns (#tokens checked, smallest top-2 gap seen)."""
    mels = pu.window_mels(o, audio, sample_rate, frontend=wb.prep_audio)
    assert len(mels) == len(wins)
    n_tok, gap_min = 0, np.inf
    for wi, row in enumerate(wins):
        if only is not None and wi not in only:
            continue
        assert len(row) >= 5 and row[:4] == [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps], (wi, row[:6])
        enc = o.forward_encoder(mels[wi])[0]
        lp = pu.teacher_forced_logprobs(o, st, enc, row)
        ok, bad, gap = pu.greedy_chain_report(lp, row, st.end_of_text, depth)
        assert ok, ("window", wi, "first wrong position", bad, "row length", len(row), "oracle top-2 gap there", gap)
        n_tok += len(row) - 4
        gap_min = min(gap_min, gap)
    return n_tok, gap_min


def test_small_10min_all_51_windows_depth_100_batch_mode():
    """Config #4 as named: `small` (V = 51 865, d = 768, 12 + 12 layers), 10 minutes of audio = 51 reference windows,
    all decoded in ONE batch-mode session (51 live rows), greedy to the reference's depth 100."""
    wl = workloads.WORKLOADS["small_10min"]
    w = wl.weights()
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(eng.dims["n_vocab"])
    audio = wl.audio()
    full, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 100)
    eng.close()
    assert len(wins) == 51
    n_tok, gap = _check_rows_are_oracle_greedy(o, st, audio, wins, 100)
    assert n_tok >= 51 * 20, n_tok                                   # the rows are long (most run to depth 100)
    assert len({tuple(r[4:]) for r in wins}) >= 45                   # and the windows decode to different text
    # the stitched stream is the fold of transcribe.rs:56-63 over exactly these rows
    assert full == wb.stitch_windows(np.array([r + [0] * (108 - len(r)) for r in wins], np.int32),
                                     np.array([len(r) for r in wins], np.int32))
    print(f"small 51 windows: {n_tok} teacher-forced decisions, smallest oracle top-2 gap {gap:.3e}")


@pytest.fixture(scope="module")
def large_v2():
    # Round 6: the variant whose logits keep the full scale 6 (log-probs up to ~75 in magnitude; only the branch gains and
    # attention strengths are depth-normalised).  Round 5's fixture also shrank the logit scale to 2.6, which makes an
    # ABSOLUTE 1e-3 easier; tests/study_large_v2_conditioning.py (CPU, 24-token top-5 walk at full window length): f32 oracle
    # vs its f64 twin 4.5e-5 on this variant (max |log-prob| 75, smallest top-2 gap 0.13) against 1.5e-5 on round 5's
    # (max |log-prob| 30) -- both well-conditioned, so 1e-3 is asserted outright here too.  Round 5's fixture stays covered
    # by the 38-row leg tests below, the `large_window` workload and tests/test_gpu_e2e.py.
    w = synth.synth_preset("large-v2", logit_depth_norm=False)
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    assert eng.dims["n_text_state"] == 1280 and eng.dims["n_text_layer"] == 32 and eng.dims["n_vocab"] == 51865
    yield eng, o
    eng.close()


def test_large_v2_10_windows_depth_100_batch_mode(large_v2):
    """Config #5's per-GPU path: large-v2 with MORE than 8 windows in the batch, so the > 8-row kernels run (split-K
    MFMA GEMMs, dec_cross_attn_stream_kernel, dec_self_attn_kernel, dec_topk_rows) -- the path the multi-window
    large-v2 figures are measured on -- greedy, depth 100, every one of the 10 rows teacher-forced."""
    eng, o = large_v2
    st = wb.SpecialTokens.for_vocab(51865)
    audio = synth.synth_audio(1900000, 1240)                         # 118.75 s -> 10 reference windows
    assert len(wb.window_extents(len(audio), 16000, WLEN)[0]) == 10
    _, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 100)
    assert len(wins) == 10
    n_tok, gap = _check_rows_are_oracle_greedy(o, st, audio, wins, 100)
    assert n_tok >= 10 * 20, n_tok
    print(f"large-v2 10 windows: {n_tok} teacher-forced decisions, smallest oracle top-2 gap {gap:.3e}")


@pytest.fixture(scope="module")
def large_v2_leg():
    """bench.py's large-v2 leg, exactly: the checkpoint without the EOT ramp and 450 s of the leg's own audio."""
    wl = workloads.WORKLOADS["large_leg"]
    w = wl.weights()
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    yield eng, o, wl.audio()
    eng.close()


def test_large_v2_leg_38_windows_tokens(large_v2_leg):
    """The configuration the bench's large-v2 figure is TIMED on (round 5's review: it had no parity check): 450 s = 38
    windows in ONE batch, i.e. 38 live rows = dec_skinny_f16x3_kernel<3> (three 16-row tiles) at d = 1280, 32 layers, greedy
    to depth 100.  First and last window against the oracle's committed LITERAL rows from raw PCM (golden `large_leg`:
    the rows bench.py checks its leg against), four windows spread over the batch teacher-forced position by position."""
    eng, o, audio = large_v2_leg
    st = wb.SpecialTokens.for_vocab(51865)
    assert len(wb.window_extents(len(audio), 16000, WLEN)[0]) == 38
    _, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 100)
    assert len(wins) == 38
    g = np.load(GOLD)
    gold = [g["large_leg_tokens"][i, :g["large_leg_lens"][i]].tolist() for i in range(2)]
    assert wins[0] == gold[0] and wins[37] == gold[1]
    n_tok, gap = _check_rows_are_oracle_greedy(o, st, audio, wins, 100, only=(0, 12, 25, 37))
    assert n_tok == 4 * 100, n_tok
    assert len({tuple(r[4:]) for r in wins}) >= 34                   # the windows decode to different text
    print(f"large-v2 leg, 38 windows: {n_tok} teacher-forced decisions, smallest oracle top-2 gap {gap:.3e}")


def test_large_v2_leg_38_rows_logprob_rows(large_v2_leg):
    """... and the same 38-row batch as a session: 16 positions, the log-prob rows of four windows spread over the batch
    (first tile, both middle tiles, the ragged last tile of 6 rows) within 1e-3 of the f32 oracle's stateless rows."""
    eng, o, audio = large_v2_leg
    st = wb.SpecialTokens.for_vocab(51865)
    res = _session_logprob_rows(eng, o, st, audio, list(range(38)), 1, 16, -1, 31, check=(0, 12, 25, 37))
    assert res["n_live"] == 38 and res["longest"] >= 16
    print(f"large-v2 38 x 1 beam: {res}")
    _assert_rows(res, "large_v2_leg_38x1_streaming")


def _session_logprob_rows(eng, o, st, audio, use_windows, max_beams, n_steps, fork_at, seed, check=None):
    """Drive a KV-cached session over `use_windows` with up to `max_beams` beams per window for `n_steps` positions and
    return the largest (and the rms over rows of the per-row largest) |session log-prob row - stateless oracle row| over
    the compared beams and steps.  Beams fork once at step `fork_at` (when max_beams > 1); every beam continues with a
    random pick among its own top-5, so rows depend on the whole history.  `check`: positions in `use_windows` whose rows are
    compared (default: all) -- every window is decoded (the batch composition is what selects the kernels), the oracle side
    (one encoder pass + one teacher-forced decoder pass per sequence, the test's cost at large-v2) runs for those only."""
    starts, lens = wb.window_extents(len(audio), 16000, WLEN)
    sess = wb.Session.begin(eng, audio, starts[use_windows], lens[use_windows], max_beams=max_beams)
    sess.set_special_mask(st.is_special)
    mels = pu.window_mels(o, audio, frontend=wb.prep_audio)
    check = set(range(len(use_windows))) if check is None else set(check)
    encs = {wdx: o.forward_encoder(mels[use_windows[wdx]])[0] for wdx in sorted(check)}
    prompt = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
    K = 5
    nw = len(use_windows)
    beams = [([prompt[0]], wdx) for wdx in range(nw)]
    parents = [-1] * nw
    records = []
    rng = np.random.default_rng(seed)
    for step in range(n_steps):
        toks = [b[0][-1] for b in beams]
        wins = [b[1] for b in beams]
        feeding = step < 3
        use_mask = (not feeding) and (step + 1 <= 5)
        ids, lps = sess.step(toks, parents, wins, apply_special_mask=use_mask, k=0 if feeding else K)
        if feeding:
            beams = [(b[0] + [prompt[step + 1]], b[1]) for b in beams]
            parents = list(range(len(beams)))
            continue
        # compare a rotating third of the rows each step (every row is covered many times; V floats per row cross PCIe)
        for slot, (seq, wdx) in enumerate(beams):
            if wdx in check and ((slot + step) % 3 == 0 or step >= n_steps - 12):
                got = sess.last_logprobs(slot)
                order = np.lexsort((np.arange(got.shape[0]), -got.astype(np.float64)))[:K]
                assert ids[slot].tolist() == order.tolist(), (step, slot)
                records.append((tuple(seq), wdx, got))
        nxt, npar = [], []
        for slot, (seq, wdx) in enumerate(beams):
            pick = int(rng.integers(0, K))
            nxt.append((seq + [int(ids[slot][pick])], wdx)); npar.append(slot)
            if max_beams > 1 and step == fork_at:
                nxt.append((seq + [int(ids[slot][(pick + 1) % K])], wdx)); npar.append(slot)
        # keep the slot order grouped by window (a window's beams are contiguous in the reference's beam list too)
        order = sorted(range(len(nxt)), key=lambda i: nxt[i][1])
        beams, parents = [nxt[i] for i in order], [npar[i] for i in order]
    n_live = len(beams)
    sess.close()
    finals = sorted({(r[0], r[1]) for r in records}, key=lambda x: -len(x[0]))
    rows = {}
    for seq, wdx in finals:
        if (seq, wdx) in rows:
            continue
        lp = pu.teacher_forced_logprobs(o, st, encs[wdx], list(seq))
        for n in range(4, len(seq) + 1):
            rows.setdefault((seq[:n], wdx), lp[n - 4])
    worst32, sq, mag = 0.0, 0.0, 0.0
    for seq, wdx, got in records:
        ref = rows[(seq, wdx)]
        fin = np.isfinite(ref)
        assert (np.isfinite(got) == fin).all()
        e = float(np.abs(got[fin] - ref[fin]).max())
        worst32 = max(worst32, e); sq += e * e
        mag = max(mag, float(np.abs(ref[fin]).max()))
    return {"hip_o32": worst32, "rms_hip_o32": (sq / max(len(records), 1)) ** 0.5, "n_live": n_live, "n_rows": len(records),
            "longest": max(len(r[0]) for r in records), "max_abs_logprob": mag}


def _assert_rows(res, name):
    """north_star: every compared log-prob row within 1e-3 of the reference-style f32 evaluation (module docstring)."""
    parity_log.record("batchmode::" + name, res["hip_o32"], LOGPROB_TOL, res["max_abs_logprob"], rms=res["rms_hip_o32"],
                      n_rows=res["n_rows"], n_live=res["n_live"], longest=res["longest"])
    assert res["n_rows"] >= 30, res
    assert res["hip_o32"] <= LOGPROB_TOL, res


def test_large_v2_batch_mode_beams_logprob_rows(large_v2):
    """large-v2, 5 windows x 2 beams = 10 live rows: batch mode with beams, i.e. dec_cross_attn_kernel<2> + the chunk
    combine at d = 1280 (the streaming kernel serves one beam per window only); 20 positions, the compared log-prob
    rows within 1e-3 of the f32 oracle's stateless rows (module docstring)."""
    eng, o = large_v2
    st = wb.SpecialTokens.for_vocab(51865)
    audio = synth.synth_audio(1900000, 1240)
    res = _session_logprob_rows(eng, o, st, audio, [0, 2, 4, 6, 9], 2, 20, 3, 11, check=(0, 2, 4))
    assert res["n_live"] == 10 and res["longest"] >= 20
    print(f"large-v2 5 x 2 beams: {res}")
    _assert_rows(res, "large_v2_5x2_beams")


def test_large_v2_batch_mode_streaming_logprob_rows(large_v2):
    """large-v2, 9 windows x 1 beam = the streaming cross-attention kernel (dec_cross_attn_stream_kernel: config #5's
    per-GPU path) at d = 1280: 24 positions, every compared row within 1e-3 of the f32 oracle's."""
    eng, o = large_v2
    st = wb.SpecialTokens.for_vocab(51865)
    audio = synth.synth_audio(1900000, 1240)
    res = _session_logprob_rows(eng, o, st, audio, list(range(9)), 1, 24, -1, 21, check=(0, 3, 5, 8))
    assert res["n_live"] == 9 and res["longest"] >= 24
    print(f"large-v2 9 x 1 beam: {res}")
    _assert_rows(res, "large_v2_9x1_streaming")


@pytest.mark.parametrize("mode", ["stream_9x1", "chunked_5x2"])
def test_small_batch_mode_session_past_the_second_self_attention_tile(mode):
    """`small` shape, batch-mode sessions for 122 positions (> 112 cached positions: the second key tile of
    dec_self_attn_kernel, the paged cache tables after re-indexing): 9 windows x 1 beam (streaming cross-attention)
    and 5 windows x 2 beams (chunked cross-attention + combine, beams forking at step 3)."""
    wl = workloads.WORKLOADS["small_10min"]
    w = synth.synth_preset("small", eot_beta=0.0)                    # no EOT ramp: no row ever ends
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(51865)
    audio = wl.audio()[:190559 * 9 + 48000]                           # 9 full windows + a 3 s tail window
    if mode == "stream_9x1":
        res = _session_logprob_rows(eng, o, st, audio, list(range(9)), 1, 122, -1, 21)
        assert res["n_live"] == 9
    else:
        res = _session_logprob_rows(eng, o, st, audio, [0, 2, 4, 6, 9], 2, 122, 3, 22)
        assert res["n_live"] == 10
    eng.close()
    assert res["longest"] >= 122
    print(f"small {mode}: {res}")
    _assert_rows(res, "small_122_positions_" + mode)
