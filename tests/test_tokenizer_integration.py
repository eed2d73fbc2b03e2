"""The tokenizer side of the boundary (src/token.rs): ids by name and the special-token mask derived exactly
as the reference derives them, from a `tokenizer.json` -- a SYNTHETIC one (no Whisper tokenizer.json exists
offline) with the real special-token names, written with the same HuggingFace `tokenizers` crate the
reference wraps.  CPU: the derivation; GPU: the derived table drives the engine."""
import json
import os

import numpy as np
import pytest

from whisper_burn_amd import synth
from whisper_burn_amd.tokens import SpecialTokens, TokenizerAdapter, special_token_name

tokenizers = pytest.importorskip("tokenizers")

N_VOCAB = 1031            # the test models' vocabulary: ids >= 1015 are special (tokens.py layout)


def write_synthetic_tokenizer_json(path):
    from tokenizers import Tokenizer, models, pre_tokenizers
    eot = N_VOCAB - 16
    vocab = {f"w{i}": i for i in range(eot)}
    vocab["<unk>"] = vocab.pop(f"w{eot - 1}")                                 # keep ids dense: the last word slot is <unk>
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    # ids eot .. eot + 15 in the layout of tokens.py: eot, sot, language(s), transcribe, notimestamps, timestamps
    names = ["<|endoftext|>", "<|startoftranscript|>", "<|en|>", "<|zh|>", "<|transcribe|>", "<|translate|>",
             "<|notimestamps|>"] + [f"<|{0.02 * i:.2f}|>" for i in range(9)]
    assert tok.add_special_tokens(names) == 16
    tok.save(path)
    return names


def test_special_tokens_are_derived_like_the_reference(tmp_path):
    path = str(tmp_path / "tokenizer.json")
    names = write_synthetic_tokenizer_json(path)
    assert json.load(open(path))["model"]["type"] == "WordLevel"
    bpe = TokenizerAdapter.from_file(path)
    assert bpe.vocab_size() == N_VOCAB
    st = bpe.special_tokens("en")
    ref = SpecialTokens.for_vocab(N_VOCAB)
    assert (st.start_of_transcript, st.language, st.transcribe, st.no_timestamps, st.end_of_text) == \
        (ref.start_of_transcript, ref.language, ref.transcribe, ref.no_timestamps, ref.end_of_text)
    assert bpe.special_token(special_token_name("language", "zh")) == ref.language + 1
    # token.rs:37-43: special <=> decode([id], skip_special = true) is empty
    assert np.array_equal(st.is_special, ref.is_special)
    assert st.is_special.sum() == 16 and not st.is_special[:N_VOCAB - 16].any()
    assert bpe.decode([3, ref.end_of_text, 5], skip_special=True) == "w3 w5"           # transcribe.rs:67
    assert names[0] in bpe.decode([3, ref.end_of_text], skip_special=False)
    with pytest.raises(KeyError):
        bpe.special_tokens("xx")


@pytest.mark.gpu
def test_tokenizer_table_drives_the_engine(tmp_path):
    import whisper_burn_amd as wb
    path = str(tmp_path / "tokenizer.json")
    write_synthetic_tokenizer_json(path)
    bpe = TokenizerAdapter.from_file(path)
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=N_VOCAB)
    eng = wb.Whisper.from_tensors(synth.synth_weights(dims, seed=4242))
    audio = synth.synth_audio(16000 * 20, 51)
    ref_tokens, _ = wb.waveform_to_tokens(eng, wb.SpecialTokens.for_vocab(N_VOCAB), audio, 16000, 5, 24)

    class Bpe:                                     # what waveform_to_text (transcribe.rs:23-29) needs from its `bpe`
        def special_tokens(self, lang):
            return bpe.special_tokens(lang)

        def decode(self, tokens, skip_special):
            return bpe.decode(tokens, skip_special)

    text, tokens = wb.waveform_to_text(eng, Bpe(), "en", audio, 16000)
    ref_tokens100, _ = wb.waveform_to_tokens(eng, wb.SpecialTokens.for_vocab(N_VOCAB), audio, 16000, 5, 100)
    assert tokens == ref_tokens100
    assert text == bpe.decode(tokens, True) and "<|" not in text and len(text.split()) >= 2
    assert ref_tokens[:4] == tokens[:4]


def test_cli_argument_errors_match_the_reference(tmp_path, capsys, monkeypatch):
    """bin/transcribe/main.rs:100-123: usage, invalid language, unreadable audio -> message on stderr, exit code 1."""
    from whisper_burn_amd import transcribe as cli
    assert cli.main(["transcribe"]) == 1
    assert "Usage: transcribe <model name> <audio file> <lang> <transcription file>" in capsys.readouterr().err
    assert cli.main(["transcribe", "m", "a.wav", "xx", "out.txt"]) == 1
    assert "Invalid language abbreviation: xx" in capsys.readouterr().err
    assert cli.main(["transcribe", "m", str(tmp_path / "missing.wav"), "en", "out.txt"]) == 1
    assert "Failed to load audio file" in capsys.readouterr().err


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path, monkeypatch):
    """The reference's CLI flow: 16 kHz mono WAV + tokenizer.json in the working directory + model -> transcript file."""
    import wave
    from whisper_burn_amd import dumpdir
    from whisper_burn_amd import transcribe as cli
    monkeypatch.chdir(tmp_path)
    write_synthetic_tokenizer_json(str(tmp_path / "tokenizer.json"))
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=N_VOCAB)
    dumpdir.write_dump_dir(synth.synth_weights(dims, seed=4242), str(tmp_path / "micro"))
    pcm = np.clip(np.round(synth.synth_audio(16000 * 6, 52) * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    assert cli.main(["transcribe", "micro", "a.wav", "en", "out.txt"]) == 0
    text = open(tmp_path / "out.txt").read()
    assert "<|" not in text and len(text.split()) >= 2


@pytest.mark.gpu
def test_cli_resamples_a_22050_hz_file_on_request(tmp_path, monkeypatch, capsys):
    """The bundled audio.wav is 22 050 Hz: the reference's CLI refuses it (main.rs:42) and so does this one, unless
    WHISPER_HIP_RESAMPLE=1 routes it through the device resampler first."""
    import wave
    import whisper_burn_amd as wb
    from whisper_burn_amd import dumpdir
    from whisper_burn_amd import transcribe as cli
    monkeypatch.chdir(tmp_path)
    write_synthetic_tokenizer_json(str(tmp_path / "tokenizer.json"))
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=N_VOCAB)
    weights = synth.synth_weights(dims, seed=4242)
    dumpdir.write_dump_dir(weights, str(tmp_path / "micro"))
    from scipy.signal import resample_poly
    hi = resample_poly(synth.synth_audio(16000 * 6, 52).astype(np.float64), 441, 320)       # a 22 050 Hz recording
    pcm = np.clip(np.round(hi * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(str(tmp_path / "a22.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(22050); w.writeframes(pcm.tobytes())
    monkeypatch.delenv("WHISPER_HIP_RESAMPLE", raising=False)
    assert cli.main(["transcribe", "micro", "a22.wav", "en", "out.txt"]) == 1
    assert "must be 16k" in capsys.readouterr().err
    monkeypatch.setenv("WHISPER_HIP_RESAMPLE", "1")
    assert cli.main(["transcribe", "micro", "a22.wav", "en", "out.txt"]) == 0
    text = open(tmp_path / "out.txt").read()
    # the same transcript as resampling by hand and calling the library
    x16 = wb.resample(pcm.astype(np.float32) / np.float32(32767.0), 22050, 16000)
    bpe = TokenizerAdapter.from_file(str(tmp_path / "tokenizer.json"))
    ref_text, _ = wb.waveform_to_text(wb.Whisper.from_tensors(weights), bpe, "en", x16, 16000)
    assert text == ref_text and len(text.split()) >= 2
