"""`python -m whisper_burn_amd.transcribe <model name> <audio file> <lang> <transcription file>`

The reference's CLI (/root/reference/src/bin/transcribe/main.rs:85-157) over the HIP engine, argument for argument:
`<model name>` is the converter's output prefix (`<name>.mpk.gz` + `<name>.cfg`, main.rs:113-126) or a dump
directory (load.rs:295), the audio must be a 16 kHz mono WAV (main.rs:41-42; 16-bit PCM is scaled by 1 / 32767 as
hound does, :44-51), `tokenizer.json` is read from the working directory (token.rs:13-19), the transcript is
written to `<transcription file>` (main.rs:151-154).  Exit code 1 with the reference's messages on bad input.

One addition: with `WHISPER_HIP_RESAMPLE=1` in the environment a mono WAV of another sample rate (the bundled
22 050 Hz audio.wav, which the reference sends through `sox`, README.md:69-74) is resampled to 16 kHz on the GPU
(wb_resample_dev) instead of being rejected.
"""
from __future__ import annotations

import os
import sys

LANGUAGES = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la "
             "mi ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be "
             "tg sd gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl mg as tt ln ha ba jw su").split()   # token.rs:50-60


def main(argv=None) -> int:
    argv = list(sys.argv if argv is None else argv)
    if len(argv) < 5:
        print(f"Usage: {argv[0]} <model name> <audio file> <lang> <transcription file>", file=sys.stderr)
        return 1
    model_name, wav_file, lang, text_file = argv[1:5]
    if lang not in LANGUAGES:
        print(f"Invalid language abbreviation: {lang}", file=sys.stderr)
        return 1
    import whisper_burn_amd as wb
    from .tokens import TokenizerAdapter

    print("Loading waveform...")
    try:
        if os.environ.get("WHISPER_HIP_RESAMPLE", "0") == "1":
            waveform, sample_rate = wb.load_audio_waveform(wav_file, any_rate=True)
            if sample_rate != 16000:
                print(f"Resampling {sample_rate} Hz -> 16000 Hz...")
                waveform, sample_rate = wb.resample(waveform, sample_rate, 16000), 16000
        else:
            waveform, sample_rate = wb.load_audio_waveform(wav_file)      # asserts 16 kHz mono like main.rs:41-42
    except Exception as e:                                                 # noqa: BLE001
        print(f"Failed to load audio file: {e}", file=sys.stderr)
        return 1
    try:
        bpe = TokenizerAdapter.from_file("tokenizer.json")
    except Exception as e:                                                 # noqa: BLE001
        print(f"Failed to load tokenizer: {e}", file=sys.stderr)
        return 1
    print("Loading model...")
    try:
        if os.path.isdir(model_name):
            whisper = wb.Whisper.load_dump_dir(model_name)
        else:
            whisper = wb.Whisper.load_burn_record(model_name + ".mpk.gz", model_name + ".cfg")
    except Exception as e:                                                 # noqa: BLE001
        print(f"Failed to load whisper model file: {e}", file=sys.stderr)
        return 1

    class Bpe:                                                             # what waveform_to_text needs (transcribe.rs:23-29)
        def special_tokens(self, language):
            return bpe.special_tokens(language)

        def decode(self, tokens, skip_special):
            return bpe.decode(tokens, skip_special)

    try:
        text, _tokens = wb.waveform_to_text(whisper, Bpe(), lang, waveform, sample_rate)
    except Exception as e:                                                 # noqa: BLE001
        print(f"Error during transcription: {e}", file=sys.stderr)
        return 1
    try:
        with open(text_file, "w") as fh:
            fh.write(text)
    except OSError as e:
        print(f"Error writing transcription file: {e}", file=sys.stderr)
        return 1
    print("Transcription finished.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
