"""Special-token table the decode driver needs (ids only; no tokenizer).

The reference looks these up by name in ./tokenizer.json
(/root/reference/src/transcribe.rs:179-185, src/token.rs:26-30, :267-295) and builds
the special-token mask by calling `is_special` on every vocab id
(transcribe.rs:243-251).  No tokenizer.json exists here, so the ids of the standard
Whisper vocabularies are tabulated from the published vocab layout; a caller that
does own a tokenizer passes its own table.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class SpecialTokens:
    start_of_transcript: int
    language: int
    transcribe: int
    no_timestamps: int
    end_of_text: int
    is_special: np.ndarray      # uint8 [V]: 1 where tokenizer.decode([id], skip_special=True) == ""

    @staticmethod
    def for_vocab(n_vocab: int, language_index: int = 0) -> "SpecialTokens":
        """Standard Whisper layouts: 51864 (.en) / 51865 (multilingual); any other size
        (synthetic test vocabularies) places the specials in the last 16 ids."""
        if n_vocab == 51864:        # gpt2 + <|endoftext|>=50256, sot=50257, 99 langs, translate ...
            eot, sot = 50256, 50257
            lang0, transcribe, notimestamps = 50258, 50358, 50362
        elif n_vocab == 51865:
            eot, sot = 50257, 50258
            lang0, transcribe, notimestamps = 50259, 50359, 50363
        else:
            assert n_vocab >= 32
            eot = n_vocab - 16
            sot, lang0, transcribe, notimestamps = eot + 1, eot + 2, eot + 4, eot + 6
            language_index = 0
        is_special = np.zeros(n_vocab, dtype=np.uint8)
        is_special[eot:] = 1
        return SpecialTokens(sot, lang0 + language_index, transcribe, notimestamps, eot, is_special)
