"""Optional mode: the greedy decoder the reference retired (commented out in mels_to_text,
/root/reference/src/transcribe.rs:314-378) and its repetition detectors (:385-447, defined but unused there),
over the KV-cached session instead of a full decoder re-run per token.

Not part of the default path: `waveform_to_text` is the live beam search (transcribe.rs:232-233, :253-312).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .model import Session, Whisper, check, special_mask_bytes
from .tokens import SpecialTokens


def _ip(tokens):
    a = np.ascontiguousarray(tokens, dtype=np.int32)
    return a, a.ctypes.data_as(_lib.c_int32_p)


def first_repetition_end(tokens: Sequence[int], period: int) -> int:
    """transcribe.rs:385-393."""
    a, p = _ip(tokens)
    r = int(_lib.load().wb_first_repetition_end(p, len(a), period))
    if r < 0:
        check(r)
    return r


def repetition_period(tokens: Sequence[int], min_repetitions: int) -> Optional[int]:
    """transcribe.rs:395-417."""
    a, p = _ip(tokens)
    r = int(_lib.load().wb_repetition_period(p, len(a), min_repetitions))
    return r if r > 0 else None


def find_repeated_tokens_index(tokens: Sequence[int], window_size: int, min_repeat_count: int) -> Optional[Tuple[int, int]]:
    """transcribe.rs:419-447."""
    a, p = _ip(tokens)
    first, end = C.c_int64(0), C.c_int64(0)
    r = _lib.load().wb_find_repeated_tokens_index(p, len(a), window_size, min_repeat_count, C.byref(first), C.byref(end))
    if r < 0:
        check(r)
    return (int(first.value), int(end.value)) if r == 1 else None


def legacy_greedy(whisper: Whisper, st: SpecialTokens, mel: np.ndarray, padding: int = 10, repeat_window_size: int = 5,
                  min_n_repeats: int = 4, max_tokens: Optional[int] = None) -> List[int]:
    """transcribe.rs:314-378 for one window's log-mel [80, T] (unclipped, unpadded: the session clips to
    n_audio_ctx - padding frames and appends the zero frames, :171-177).  Argmax of the unmasked row, stop on
    exp(eot_logit - token_logit) > 0.5 (:351), on the repetition detector (:369-377) or at n_text_ctx tokens (:317-320).
    log-softmax rows stand in for the raw logits: both criteria only use differences within a row."""
    n_ctx = whisper.decoder_ctx_size() if max_tokens is None else min(max_tokens, whisper.decoder_ctx_size())
    tokens = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
    sess = Session.begin_mel(whisper, [np.asarray(mel, np.float32).reshape(80, -1)], max_beams=1, padding=padding)
    try:
        for i, t in enumerate(tokens[:-1]):                               # the prompt only feeds the KV cache
            sess.step([t], [-1 if i == 0 else 0], [0], apply_special_mask=False, k=0)
        while True:
            if len(tokens) >= n_ctx:
                tokens.append(st.end_of_text)
                break
            ids, lps = sess.step([tokens[-1]], [0], [0], apply_special_mask=False, k=1)
            token_id, token_lp = int(ids[0][0]), float(lps[0][0])
            eot_lp = float(sess.last_logprobs(0)[st.end_of_text])
            tokens.append(token_id)
            if math.exp(eot_lp - token_lp) > 0.5:
                if token_id != st.end_of_text:
                    tokens.append(st.end_of_text)
                break
            hit = find_repeated_tokens_index(tokens, repeat_window_size, min_n_repeats)
            if hit is not None:
                tokens = tokens[:hit[1]]
                tokens.append(st.end_of_text)
                break
    finally:
        sess.close()
    return tokens


def waveform_to_tokens_prompted(whisper: Whisper, st: SpecialTokens, waveform, sample_rate: int = 16000,
                                start_of_prev: Optional[int] = None, n_prev_tokens: int = 5, beam_size: int = 5,
                                max_depth: int = 100):
    """waveform_to_text (transcribe.rs:23-74) with the prompt conditioning the reference wrote and then disabled
    (:43-50, :188-199; shadowed at :201, "including the prev tokens causes whisper to hallucinate", :186): every
    window after the first starts from [<|startofprev|>, the last five non-special tokens so far, <|startoftranscript|>,
    language, <|transcribe|>, <|notimestamps|>].  Returns (stitched ids, per-window rows with their prompts).
    Windows depend on their predecessors: they are decoded one at a time (no batching, no sharding)."""
    from .model import _f32, _fp, decode_params, max_waveform_samples, window_extents
    lib = _lib.load()
    wav = _f32(waveform).reshape(-1)
    p = decode_params(st, beam_size, max_depth)
    sop = st.start_of_prev if start_of_prev is None else start_of_prev
    wlen = max_waveform_samples(whisper.max_mel_frames() - p.padding)
    n_win = len(window_extents(len(wav), sample_rate, wlen, p.overlap_seconds)[0])
    stride = 1 + n_prev_tokens + 4 + max_depth + 4
    rows = np.zeros((max(n_win, 1), stride), np.int32)
    lens = np.zeros(max(n_win, 1), np.int32)
    stitched = np.zeros(max(n_win, 1) * stride, np.int32)
    n_st = C.c_int64(0)
    mask = special_mask_bytes(whisper, st.is_special)
    check(lib.wb_waveform_to_tokens_prompted(whisper._h, _fp(wav), len(wav), sample_rate, C.byref(p),
                                             mask.ctypes.data_as(_lib.c_uint8_p), int(sop), n_prev_tokens,
                                             rows.ctypes.data_as(_lib.c_int32_p), stride,
                                             lens.ctypes.data_as(_lib.c_int32_p),
                                             stitched.ctypes.data_as(_lib.c_int32_p), len(stitched), C.byref(n_st)))
    return stitched[:n_st.value].tolist(), [rows[i, :lens[i]].tolist() for i in range(n_win)]
