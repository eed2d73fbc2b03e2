"""Oracle (test infrastructure): restatement of /root/reference/src/transcribe.rs.

Windowing, the decode driver (`mels_to_text` with the live beam search,
transcribe.rs:148-312) and the token-overlap stitch. The tokenizer is out of
scope: the caller supplies the special-token ids and the `is_special` mask the
reference derives from `tokenizer.json` (transcribe.rs:179-185, :243-251), and
token ids are returned instead of text.

The algorithm is restated AS WRITTEN: no KV cache, the whole decoder re-run over
the whole prefix for every beam at every step, cross-attention K/V recomputed
per layer per call (transcribe.rs:270, mod.rs:482-490), logits for all positions.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import beam as obeam
from .mel import max_waveform_samples, prep_audio
from .model import OracleWhisper, log_softmax


@dataclass
class SpecialTokens:
    """Ids the reference looks up by name (transcribe.rs:179-185) + the vocab mask."""
    start_of_transcript: int
    language: int
    transcribe: int
    no_timestamps: int
    end_of_text: int
    is_special: np.ndarray          # bool [V], transcribe.rs:243-244


def find_chunk_overlap(prev_tokens: Sequence[int], curr_tokens: Sequence[int],
                       max_n_offsets: int, min_n_overlaps: int) -> Optional[Tuple[int, int]]:
    """transcribe.rs:76-110."""
    max_overlap = 0
    max_overlap_indices = (0, 0)
    n_offsets = min(len(prev_tokens), len(curr_tokens), max_n_offsets)
    for offset in range(n_offsets):
        prev_start_index = len(prev_tokens) - 1 - offset
        matches = [i for i, (old, new) in enumerate(zip(prev_tokens[prev_start_index:], curr_tokens))
                   if old == new]
        n_overlap = len(matches)
        if n_overlap > max_overlap:
            max_overlap = n_overlap
            curr_overlap_index = matches[0]
            max_overlap_indices = (prev_start_index + curr_overlap_index, curr_overlap_index)
    return max_overlap_indices if max_overlap >= min_n_overlaps else None


def window_extents(n_samples: int, sample_rate: int, window_length_samples: int) -> List[Tuple[int, int]]:
    """transcribe.rs:120-128: [start, end) of every window."""
    chunk_overlap = sample_rate * 3
    shift = max(max(window_length_samples - chunk_overlap, 0), 1)
    iter_len = max(n_samples - 1, 0) // shift + 1
    return [(i * shift, min(i * shift + window_length_samples, n_samples)) for i in range(iter_len)]


def mels_to_tokens(whisper: OracleWhisper, st: SpecialTokens, mels: torch.Tensor, padding: int = 10,
                   beam_size: int = 5, max_depth: int = 100, prev_nonspecial_tokens: Sequence[int] = (),
                   start_of_prev: Optional[int] = None) -> List[int]:
    """transcribe.rs:148-312 (token ids instead of text). mels: [1, 80, T].

    `start_of_prev` (None = the live code): the retired prompt conditioning of :188-199 -- the reference builds
    `[start_of_prev, prev tokens...]` and then shadows it with an empty list (:201).  With an id given, that first
    list is kept, exactly as :195-199 computes it."""
    n_ctx_max_encoder = whisper.encoder_ctx_size()
    _, n_mel, n_ctx = mels.shape
    mels = torch.cat([mels[0:1, :, 0:min(n_ctx, n_ctx_max_encoder - padding)],
                      torch.zeros(1, n_mel, padding, dtype=mels.dtype)], 2)      # :171-177
    initial_tokens: List[int] = []                                               # :201
    if start_of_prev is not None and len(prev_nonspecial_tokens) > 0:            # :195-199 (dead in the reference)
        initial_tokens = [start_of_prev] + list(prev_nonspecial_tokens)
    initial_tokens += [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]   # :203
    initial = obeam.BeamNode(seq=[(t, 0.0) for t in initial_tokens], log_prob=0.0)             # :205-220
    encoder_output = whisper.forward_encoder(mels)                               # :222
    neg_inf = float("-inf")
    maskout = torch.tensor(np.where(st.is_special, neg_inf, 0.0), dtype=torch.float32)  # :244

    def is_finished(seq) -> bool:                                                # :235-241
        return bool(seq) and seq[-1][0] == st.end_of_text

    def next_lp(beams):
        """transcribe.rs:253-284: per beam the f32 log-prob row at its last token."""
        max_seq_len = max((len(b.seq) for b in beams), default=0)
        toks = [[t for t, _ in b.seq] + [0] * (max_seq_len - len(b.seq)) for b in beams]
        logits = whisper.forward_decoder(torch.tensor(toks, dtype=torch.long),
                                         encoder_output.repeat(len(beams), 1, 1))
        if not max_seq_len > 5:
            logits = logits + maskout[None, None]
        log_probs = log_softmax(logits, 2)
        return [log_probs[i, len(b.seq) - 1].numpy().astype(np.float64) for i, b in enumerate(beams)]

    def step(beams, _next, is_fin, k):
        """beam.rs:39-79 with the V-wide scans done by `top_indices_fast`."""
        rows = next_lp(beams)
        finished, new_beams = [], []
        for node, lp in zip(beams, rows):
            if is_fin(node.seq):
                finished.append(node)
            else:
                scores = node.log_prob + lp                                      # :299
                for tok in obeam.top_indices_fast(scores, k):
                    new_beams.append(obeam.BeamNode(seq=node.seq + [(int(tok), float(lp[tok]))],
                                                    log_prob=float(scores[tok])))
        return obeam.get_top_elements(new_beams, lambda b: b.log_prob, k) + \
            obeam.get_top_elements(finished, lambda b: b.log_prob, k)

    seq = obeam.beam_search([initial], None, is_finished, beam_size, max_depth, step_fn=step)  # :309
    return [t for t, _ in seq]


def waveform_to_tokens(whisper: OracleWhisper, st: SpecialTokens, waveform: np.ndarray,
                       sample_rate: int = 16000, beam_size: int = 5, max_depth: int = 100,
                       return_windows: bool = False, start_of_prev: Optional[int] = None):
    """transcribe.rs:23-74 without the tokenizer: returns the stitched token ids.
    `start_of_prev`: see mels_to_tokens (None = the reference's live behaviour)."""
    padding = 10
    n_per_window = max_waveform_samples(whisper.encoder_ctx_size() - padding)     # :32-34
    tokens: List[int] = []
    per_window = []
    wav = torch.as_tensor(np.asarray(waveform, dtype=np.float32))
    for start, end in window_extents(len(wav), sample_rate, n_per_window):
        mel = prep_audio(wav[start:end][None], float(sample_rate))                # :134
        prev_normal_tokens = [t for t in reversed(tokens) if not st.is_special[t]][:5][::-1]   # :43-50
        new_tokens = mels_to_tokens(whisper, st, mel, padding, beam_size, max_depth, prev_normal_tokens, start_of_prev)
        per_window.append(list(new_tokens))
        tokens = stitch(tokens, new_tokens)
    return (tokens, per_window) if return_windows else tokens


def stitch(tokens: List[int], new_tokens: List[int]) -> List[int]:
    """transcribe.rs:56-63."""
    ov = find_chunk_overlap(tokens, new_tokens, 40, 3)
    if ov is not None:
        prev_index, curr_index = ov
        return tokens[:prev_index] + list(new_tokens[curr_index:])
    return tokens + list(new_tokens)


# ---- the reference's retired greedy decoder (dead code there; an optional mode here) ---------------------

def first_repetition_end(tokens: Sequence[int], period: int) -> int:
    """transcribe.rs:385-393.  (usize arithmetic: len < period would panic in the reference -> ValueError.)"""
    n = len(tokens)
    if n < period:
        raise ValueError("tokens.len() - period underflows")
    for i in reversed(range(period, n - period)):
        if list(tokens[i - period:i]) != list(tokens[i:i + period]):
            return i + 1
    return period


def repetition_period(tokens: Sequence[int], min_repetitions: int) -> Optional[int]:
    """transcribe.rs:395-417."""
    n = len(tokens)
    for i in reversed(range(n)):
        period = n - i
        if i // period < min_repetitions:
            return None
        ok = True
        for j in range(min_repetitions):
            e = i - period * j
            s = e - period
            if list(tokens[s:e]) != list(tokens[i:i + period]):
                ok = False
                break
        if ok:
            return period
    return None


def find_repeated_tokens_index(tokens: Sequence[int], window_size: int, min_repeat_count: int) -> Optional[Tuple[int, int]]:
    """transcribe.rs:419-447: (index of the first window equal to the last one, index of the second)."""
    n = len(tokens)
    if 2 * window_size > n:
        return None                                              # :425-427
    last_index = n - window_size
    last_window = list(tokens[last_index:])
    repeats = [i for i in range(0, last_index - window_size + 1)          # 0..=(last_index - window_size), :432
               if list(tokens[i:i + window_size]) == last_window]
    if len(repeats) >= min_repeat_count:                          # :439-443 (min_repeat_count >= 2 or .unwrap() panics)
        return repeats[0], repeats[1]
    return None


def legacy_greedy(whisper: OracleWhisper, st: SpecialTokens, mels: torch.Tensor, padding: int = 10,
                  repeat_window_size: int = 5, min_n_repeats: int = 4, max_tokens: Optional[int] = None) -> List[int]:
    """The commented-out loop of mels_to_text (transcribe.rs:314-378): argmax of the RAW last-row logits (no special-token
    mask, no log-softmax), stop when exp(eot_logit - token_logit) > 0.5 (:351) or when the last 5 tokens already
    occurred >= 4 times (:369-377, truncating to the second occurrence), or at n_text_ctx tokens (:317-320).
    `max_tokens` (not in the reference) bounds the loop for tests."""
    n_ctx_max_encoder = whisper.encoder_ctx_size()
    n_mel = mels.shape[2]
    mels = torch.cat([mels[:, :, :min(n_mel, n_ctx_max_encoder - padding)], torch.zeros(1, mels.shape[1], padding)], 2)
    enc = whisper.forward_encoder(mels)
    n_ctx = whisper.decoder_ctx_size() if max_tokens is None else min(max_tokens, whisper.decoder_ctx_size())
    tokens = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
    while True:
        if len(tokens) >= n_ctx:
            tokens.append(st.end_of_text)
            break
        out = whisper.forward_decoder(torch.tensor([tokens], dtype=torch.long), enc)
        last = out[0, len(tokens) - 1]
        token_id = int(torch.argmax(last))                       # burn argmax: first maximum
        token_logit = float(last[token_id])
        eot_logit = float(last[st.end_of_text])
        tokens.append(token_id)
        if math.exp(eot_logit - token_logit) > 0.5:
            if token_id != st.end_of_text:
                tokens.append(st.end_of_text)
            break
        hit = find_repeated_tokens_index(tokens, repeat_window_size, min_n_repeats)
        if hit is not None:
            tokens = tokens[:hit[1]]
            tokens.append(st.end_of_text)
            break
    return tokens
