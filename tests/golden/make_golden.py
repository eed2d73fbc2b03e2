#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container, where /root/reference exists).

The reference ships no golden vectors and cannot be run here, so the
goldens are (a) the reference's one data fixture, `audio.wav`, brought to the 16 kHz mono s16 form
its CLI requires (README.md:69-74 tells the user to run `sox`; sox is absent -> polyphase
resampling 320/441), and (b) frozen outputs of the CPU oracle on it -- they pin the oracle against
drift (PyTorch / NumPy upgrades) and give the GPU tests vectors that do not depend on running the
oracle.  Usage: python tests/golden/make_golden.py
"""
import os
import sys
import wave

import numpy as np
import torch
from scipy.signal import resample_poly

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd")]

from oracle import mel as omel                      # noqa: E402
from oracle import transcribe as otr                # noqa: E402
from oracle.model import OracleWhisper, log_softmax  # noqa: E402
from whisper_burn_amd import synth                  # noqa: E402
from whisper_burn_amd.tokens import SpecialTokens   # noqa: E402


def load_reference_wav(path="/root/reference/audio.wav"):
    with wave.open(path, "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2
        sr = w.getframerate()
        x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    # src/bin/transcribe/main.rs:44-51: s / (2^(bits-1) - 1)
    xf = x.astype(np.float64) / 32767.0
    y = resample_poly(xf, 320, 441) if sr == 22050 else xf
    return np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)


def ost(st):
    return otr.SpecialTokens(st.start_of_transcript, st.language, st.transcribe, st.no_timestamps,
                             st.end_of_text, st.is_special.astype(bool))


def main():
    pcm16 = load_reference_wav()
    assert pcm16.shape[0] == 122276, pcm16.shape
    np.savez_compressed(os.path.join(HERE, "audio_16k_s16.npz"), pcm=pcm16)
    audio = pcm16.astype(np.float32) / np.float32(32767.0)

    out = {}
    mel = omel.prep_audio(torch.from_numpy(audio)[None])[0].numpy()          # [80, 764]
    out["mel_shape"] = np.array(mel.shape)
    out["mel_head"] = mel[:, :160].copy()                                     # first 1.6 s, all rows
    out["mel_strided"] = mel[::4, ::9].copy()

    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    wts = synth.synth_weights(dims, seed=4242)
    o = OracleWhisper(wts)
    st = SpecialTokens.for_vocab(1031)
    melp = torch.cat([torch.from_numpy(mel)[None], torch.zeros(1, 80, 10)], 2)
    enc = o.forward_encoder(melp)
    out["micro_enc_shape"] = np.array(enc.shape)
    out["micro_enc_strided"] = enc[0, ::6, ::3].numpy().copy()
    prefix = torch.tensor([[st.start_of_transcript, st.language, st.transcribe, st.no_timestamps, 17, 400, 3]])
    lp = log_softmax(o.forward_decoder(prefix, enc)[0], 1)
    v, ix = torch.topk(lp, 8, dim=1)
    out["micro_prefix"] = prefix.numpy()
    out["micro_top_lp"] = v.numpy().copy()
    out["micro_top_id"] = ix.numpy().astype(np.int32)
    out["micro_greedy"] = np.array(otr.waveform_to_tokens(o, ost(st), audio, 16000, 1, 24), np.int32)
    out["micro_beam5"] = np.array(otr.waveform_to_tokens(o, ost(st), audio, 16000, 5, 24), np.int32)

    ot = OracleWhisper(synth.synth_preset("tiny.en"))
    stt = SpecialTokens.for_vocab(51864)
    out["tiny_en_greedy"] = np.array(otr.waveform_to_tokens(ot, ost(stt), audio, 16000, 1, 16), np.int32)
    out["tiny_en_beam5"] = np.array(otr.waveform_to_tokens(ot, ost(stt), audio, 16000, 5, 8), np.int32)
    np.savez_compressed(os.path.join(HERE, "oracle_outputs.npz"), **out)
    for k, a in out.items():
        print(k, a.shape, a.dtype)


if __name__ == "__main__":
    main()
