"""Special-token table the decode driver needs (ids only; the tokenizer stays on the caller's side).

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
    start_of_prev: int = -1     # <|startofprev|> (transcribe.rs:181); only the optional prompt-conditioning mode uses it

    @staticmethod
    def for_vocab(n_vocab: int, language_index: int = 0) -> "SpecialTokens":
        """Standard Whisper layouts: 51864 (.en) / 51865 (multilingual); any other size
        (synthetic test vocabularies) places the specials in the last 16 ids."""
        if n_vocab == 51864:        # gpt2 + <|endoftext|>=50256, sot=50257, 99 langs, translate ...
            eot, sot = 50256, 50257
            lang0, transcribe, notimestamps = 50258, 50358, 50362
            sop = 50360
        elif n_vocab == 51865:
            eot, sot = 50257, 50258
            lang0, transcribe, notimestamps = 50259, 50359, 50363
            sop = 50361
        else:
            assert n_vocab >= 32
            eot = n_vocab - 16
            sot, lang0, transcribe, notimestamps = eot + 1, eot + 2, eot + 4, eot + 6
            sop = eot + 5
            language_index = 0
        is_special = np.zeros(n_vocab, dtype=np.uint8)
        is_special[eot:] = 1
        return SpecialTokens(sot, lang0 + language_index, transcribe, notimestamps, eot, is_special, sop)


# ---- tokenizer integration (src/token.rs) -------------------------------------------------------------------

def special_token_name(kind: str, language: str = "en") -> str:
    """`SpecialToken::to_string` (token.rs:280-295)."""
    return {"endoftext": "<|endoftext|>", "startoftranscript": "<|startoftranscript|>", "translate": "<|translate|>",
            "transcribe": "<|transcribe|>", "startoflm": "<|startoflm|>", "startofprev": "<|startofprev|>",
            "nospeech": "<|nospeech|>", "notimestamps": "<|notimestamps|>", "language": f"<|{language}|>"}[kind]


class TokenizerAdapter:
    """The calls transcribe.rs makes on `Gpt2Tokenizer` (token.rs:12-48), over a HuggingFace `tokenizers.Tokenizer`
    (the crate the reference wraps, Cargo.lock:3504-3505) loaded from a `tokenizer.json`."""

    def __init__(self, tokenizer):
        self.tok = tokenizer

    @staticmethod
    def from_file(path: str = "tokenizer.json") -> "TokenizerAdapter":      # token.rs:13-19
        import tokenizers
        return TokenizerAdapter(tokenizers.Tokenizer.from_file(path))

    def special_token(self, name: str):                                     # token.rs:26-30
        return self.tok.token_to_id(name)

    def decode(self, tokens, skip_special: bool = True) -> str:             # token.rs:32-35
        return self.tok.decode([int(t) for t in tokens], skip_special_tokens=skip_special)

    def is_special(self, token: int) -> bool:                               # token.rs:37-43
        try:
            return self.tok.decode([int(token)], skip_special_tokens=True) == ""
        except Exception:
            return False

    def vocab_size(self) -> int:                                            # token.rs:45-47
        return self.tok.get_vocab_size(with_added_tokens=True)

    def special_tokens(self, language: str = "en") -> SpecialTokens:
        """The five ids transcribe.rs:179-185 looks up and the mask transcribe.rs:243-251 builds by decoding every
        vocabulary id -- built ONCE here (the reference rebuilds it for every window)."""
        ids = {}
        for kind in ("startoftranscript", "language", "transcribe", "notimestamps", "endoftext"):
            v = self.special_token(special_token_name(kind, language))
            if v is None:
                raise KeyError(f"tokenizer has no {special_token_name(kind, language)}")
            ids[kind] = int(v)
        mask = np.array([1 if self.is_special(t) else 0 for t in range(self.vocab_size())], dtype=np.uint8)
        sop = self.special_token(special_token_name("startofprev"))
        return SpecialTokens(ids["startoftranscript"], ids["language"], ids["transcribe"], ids["notimestamps"],
                             ids["endoftext"], mask, -1 if sop is None else int(sop))
