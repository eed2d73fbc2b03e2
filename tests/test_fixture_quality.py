"""CPU: the synthetic fixtures are NOT degenerate, and the committed golden rows are the oracle's.

Round 1's random-init weights decoded every window to one or two repeated tokens, so "token-exact" checked
one decision per window.  These tests pin the properties the parity evidence rests on, on the oracle's own
output, so the fixtures cannot silently collapse again:

  * every greedy window of the benchmarked workload has >= 50 % distinct tokens, the windows differ from each
    other, at least one window ends on <|endoftext|> between depth 10 and 90 and at least one runs to max_depth;
  * the committed rows are exactly what the oracle decodes (literal loop for the bench workload; teacher-forced
    one-pass check -- see parity_util -- for every greedy workload), and the smallest top-2 log-prob gap over
    every decision is far above fp32 round-off, so a token mismatch on the GPU is a defect, not a tie.
"""
import os

import numpy as np
import pytest

import parity_util as pu
import workloads
from oracle import transcribe as otr
from oracle.model import OracleWhisper
from whisper_burn_amd.tokens import SpecialTokens

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_outputs.npz")
MIN_GAP = 2e-3            # smallest tolerated top-2 gap of a golden greedy decision (log-prob units)


def rows_of(name):
    g = np.load(GOLD)
    t, n = g[f"{name}_tokens"], g[f"{name}_lens"]
    return [t[i, :n[i]].tolist() for i in range(len(n))]


def test_bench_workload_is_not_degenerate():
    wl = workloads.WORKLOADS["tiny_bench"]
    rows = rows_of("tiny_bench")
    st = SpecialTokens.for_vocab(51864)
    assert len(rows) == 3
    for r in rows:
        assert r[:4] == [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
        assert pu.diversity(r) >= 0.5, (len(r), pu.diversity(r))
    gen = [r[4:] for r in rows]
    for a in range(3):
        for b in range(a + 1, 3):       # windows differ: the audio decides, not the position
            n = min(len(gen[a]), len(gen[b]))
            assert sum(x == y for x, y in zip(gen[a], gen[b])) <= n // 4
    depths = [len(g) for g in gen]
    assert any(g[-1] == st.end_of_text and 10 <= len(g) <= 90 for g in gen), depths
    assert any(len(g) == wl.depth and g[-1] != st.end_of_text for g in gen), depths   # one window runs to max_depth


def test_every_preset_decodes_diverse_tokens():
    for name in ("small_10min", "large_window", "base_beam5", "tiny_beam5"):
        for r in rows_of(name):
            assert pu.diversity(r) >= 0.5, (name, r)
    assert rows_of("small_10min")[0][4:] != rows_of("small_10min")[1][4:]
    assert all(len(r) == 4 + 100 for r in rows_of("tiny_beam5"))       # the reference's live setting runs to depth 100
    assert all(len(r) == 4 + 32 for r in rows_of("base_beam5"))
    assert any(len(r) < 4 + 100 for r in rows_of("base_beam5_eot"))    # ... and beams that finish on EOT


@pytest.mark.parametrize("name", ["tiny_bench", "small_10min", "tiny_whisper30"])
def test_golden_rows_are_the_oracles_greedy_chain(name):
    wl = workloads.WORKLOADS[name]
    o = OracleWhisper(wl.weights(), frame_limit_x2=wl.frame_limit_x2)
    st = SpecialTokens.for_vocab(o.dims.n_vocab)
    mels = pu.window_mels(o, wl.audio())
    sel = range(len(mels)) if wl.windows is None else wl.windows
    for row, wi in zip(rows_of(name), sel):
        enc = o.forward_encoder(mels[wi])[0]
        lp = pu.teacher_forced_logprobs(o, st, enc, row)
        ok, bad, gap = pu.greedy_chain_report(lp, row, st.end_of_text, wl.depth)
        assert ok, (name, wi, bad)
        assert gap >= MIN_GAP, (name, wi, gap)


def test_bench_workload_literal_oracle_matches_golden():
    """The literal loop (full-prefix re-run per step) on exactly bench.py's workload."""
    wl = workloads.WORKLOADS["tiny_bench"]
    o = OracleWhisper(wl.weights())
    st = SpecialTokens.for_vocab(51864)
    toks, wins = otr.waveform_to_tokens(o, pu.ost(st), wl.audio(), 16000, wl.beam, wl.depth, return_windows=True)
    assert wins == rows_of("tiny_bench")
    g = np.load(GOLD)
    assert toks == g["tiny_bench_stitched"].tolist()


def test_deep_fixtures_are_well_conditioned():
    """Round 5's depth normalisation (synth.synth_weights): at `small` (12 + 12 layers, full 14.9 s window) the f32 oracle sits
    within 2e-4 of the f64 evaluation of the same operators over a 40-token top-5 walk (measured 4e-5 - 8e-5; before the
    normalisation 7.8e-4, and 2e-2 at large-v2, where two correct f32 evaluations could not be held to the north star's 1e-3).
    This is the property that lets the GPU tests assert |hip - oracle_f32| <= 1e-3 outright at every size; large-v2 itself is
    too heavy for the CPU suite (its walk is recorded in LABLOG R5.1: 1.3e-5 over 101 rows)."""
    import torch
    from oracle import mel as omel
    from oracle.model import log_softmax
    from whisper_burn_amd import synth
    w = synth.synth_preset("small", eot_beta=0.0)
    o32, o64 = OracleWhisper(w), OracleWhisper(w, dtype=torch.float64)
    st = SpecialTokens.for_vocab(o32.dims.n_vocab)
    audio = synth.synth_audio(160 * 1490 + 100, 1240)
    mel = omel.prep_audio(torch.from_numpy(audio[:160 * 1490])[None], 16000.0)
    mel = torch.cat([mel, torch.zeros(1, 80, 10)], 2)
    xa32, xa64 = o32.forward_encoder(mel), o64.forward_encoder(mel)
    assert float((xa32.double() - xa64).abs().max()) < 5e-5
    maskv = torch.tensor(np.where(np.asarray(st.is_special).astype(bool), -np.inf, 0.0))
    seq = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
    rng = np.random.default_rng(5)
    L = 44
    while len(seq) < L:
        lg = o32.forward_decoder(torch.tensor([seq]), xa32)[0, -1].double()
        if len(seq) <= 5:
            lg = lg + maskv
        seq.append(int(torch.topk(lg, 5).indices[int(rng.integers(0, 5))]))
    toks = torch.tensor([seq])
    l32, l64 = o32.forward_decoder(toks, xa32)[0], o64.forward_decoder(toks, xa64)[0]
    worst = 0.0
    for p in range(3, L):
        m = maskv if p + 1 <= 5 else 0.0
        a, b = log_softmax(l32[p].double() + m, 0), log_softmax(l64[p] + m, 0)
        fin = torch.isfinite(b)
        worst = max(worst, float((a[fin] - b[fin]).abs().max()))
    assert worst <= 2e-4, worst
    assert len(set(seq[4:])) >= (L - 4) // 2          # ... and the walk is not a fixed point
