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
