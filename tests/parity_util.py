"""Test helpers (checker side): oracle plumbing shared by the CPU and GPU parity tests.

`teacher_forced_*`: the reference's decode loop re-runs the stateless decoder over the whole prefix at
every step (transcribe.rs:253-307).  Because the decoder is causal (mod.rs:152, :535-544), ONE stateless
forward over a finished sequence yields, at position p, exactly the logits the loop saw when the prefix
had p + 1 tokens -- so a greedy token sequence is the oracle's own greedy output iff every position's
masked-log-softmax argmax (lowest id on ties, beam.rs:81-110 with k = 1) equals the next token.  That
check costs one forward instead of max_depth of them, which is what makes depth-100 parity at real
model shapes affordable; the literal loop (oracle.transcribe) is still run wherever it is cheap.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import mel as omel
from oracle import transcribe as otr
from oracle.model import OracleWhisper, log_softmax


def ost(st) -> otr.SpecialTokens:
    return otr.SpecialTokens(st.start_of_transcript, st.language, st.transcribe, st.no_timestamps,
                             st.end_of_text, np.asarray(st.is_special).astype(bool))


def window_mels(oracle: OracleWhisper, audio: np.ndarray, sample_rate: int = 16000, padding: int = 10, frontend=None):
    """Per reference window: the clipped + zero-padded log-mel the decode driver feeds the encoder
    (transcribe.rs:120-128, :134, :171-177).

    `frontend` (window f32 [N] -> log-mel [1, 80, N // 160]) replaces the oracle's own prep_audio: the model-parity
    tests at 1e-3 hand BOTH sides the same log-mel, because the oracle's f32 dense DFT is itself 2e-5 away from the
    exact (f64) log-mel -- 3x further than the HIP frontend -- and the synthetic checkpoints amplify a log-mel
    difference ~25x into the logits (tools/diag_stage_error.py: 5.7e-4 of logit difference from the oracle's mel
    rounding alone, against 6e-5 for each side's whole decoder).  The frontend has its own tests and tolerance."""
    n_ctx = oracle.encoder_ctx_size()
    wlen = omel.max_waveform_samples(n_ctx - padding)
    out = []
    for start, end in otr.window_extents(len(audio), sample_rate, wlen):
        win = np.asarray(audio[start:end], np.float32)
        if frontend is not None:
            mel = torch.from_numpy(np.asarray(frontend(win), np.float32).reshape(1, 80, -1))
        else:
            mel = omel.prep_audio(torch.from_numpy(win)[None], float(sample_rate))
        mel = torch.cat([mel[:, :, :min(mel.shape[2], n_ctx - padding)], torch.zeros(1, 80, padding)], 2)
        out.append(mel)
    return out


def teacher_forced_logprobs(oracle: OracleWhisper, st, enc: torch.Tensor, seq, mask_until_len: int = 5) -> np.ndarray:
    """Rows p = 3 .. len(seq) - 1 of log_softmax(masked logits) from ONE stateless forward: row p is what
    transcribe.rs:276-284 reads when the beam holds seq[:p + 1].  enc: [C, d].  Returns [len(seq) - 3, V]."""
    toks = torch.tensor([list(seq)], dtype=torch.long)
    logits = oracle.forward_decoder(toks, enc[None])[0]
    maskv = torch.tensor(np.where(np.asarray(st.is_special).astype(bool), -np.inf, 0.0), dtype=torch.float32)
    rows = []
    for p in range(3, len(seq)):
        lg = logits[p]
        if p + 1 <= mask_until_len:          # max_seq_len <= 5  (transcribe.rs:271-275)
            lg = lg + maskv
        rows.append(log_softmax(lg, 0).numpy())
    return np.stack(rows)


def greedy_chain_report(lp_rows: np.ndarray, seq, eot: int, max_depth: int):
    """Check a token sequence against teacher-forced rows: returns (ok, first_bad_position, min_top2_gap).
    ok: every generated token is the argmax (lowest id on ties) of its row, and the sequence stops exactly
    where beam.rs:22-31 with k = 1 stops (EOT chosen, or max_depth tokens)."""
    seq = list(seq)
    n_gen = len(seq) - 4
    min_gap = np.inf
    for i in range(n_gen):
        row = lp_rows[i]
        best = int(np.flatnonzero(row == row.max())[0])
        top2 = np.partition(row[np.isfinite(row)], -2)[-2:]
        min_gap = min(min_gap, float(top2[1] - top2[0]))
        if best != seq[4 + i]:
            return False, 4 + i, min_gap
    ended = (n_gen == max_depth) or (n_gen > 0 and seq[-1] == eot)
    no_early_eot = all(t != eot for t in seq[4:-1])
    return bool(ended and no_early_eot), -1, min_gap


def diversity(seq) -> float:
    gen = list(seq)[4:]
    return len(set(gen)) / max(1, len(gen))
