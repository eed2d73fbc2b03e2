#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container, where /root/reference exists).

The reference ships no golden vectors and cannot be run here, so the
goldens are (a) the reference's one data fixture, `audio.wav`, brought to the 16 kHz mono s16 form
its CLI requires (README.md:69-74 tells the user to run `sox`; sox is absent -> polyphase
resampling 320/441), and (b) frozen outputs of the CPU oracle -- the LITERAL restatement of the
reference's decode loop (oracle/transcribe.py: full-prefix decoder re-run per step, no KV cache) -- on
it and on the seeded synthetic workloads of BASELINE.json's configs #2-#5.  They pin the oracle against
drift (PyTorch / NumPy upgrades) and give the GPU tests vectors that do not depend on running the
oracle on the GPU box.

Usage: python tests/golden/make_golden.py [section ...]     (sections: wav micro tiny base small large)
Existing entries of other sections are kept.
"""
import os
import sys
import time
import wave

import numpy as np
import torch
from scipy.signal import resample_poly

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd"), os.path.join(ROOT, "tests")]

from oracle import mel as omel                      # noqa: E402
from oracle import transcribe as otr                # noqa: E402
from oracle.model import OracleWhisper, log_softmax  # noqa: E402
from whisper_burn_amd import synth                  # noqa: E402
from whisper_burn_amd.tokens import SpecialTokens   # noqa: E402
import parity_util as pu                            # noqa: E402
import workloads                                    # noqa: E402

OUT = os.path.join(HERE, "oracle_outputs.npz")


def load_reference_wav(path="/root/reference/audio.wav"):
    with wave.open(path, "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2
        sr = w.getframerate()
        x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    # src/bin/transcribe/main.rs:44-51: s / (2^(bits-1) - 1)
    xf = x.astype(np.float64) / 32767.0
    y = resample_poly(xf, 320, 441) if sr == 22050 else xf
    return np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)


def pack(rows):
    """list of token lists -> (int32 [n, max], int32 lens)"""
    n = max(len(r) for r in rows)
    a = np.zeros((len(rows), n), np.int32)
    for i, r in enumerate(rows):
        a[i, :len(r)] = r
    return a, np.array([len(r) for r in rows], np.int32)


def decode_windows(o, st, audio, beam, depth, windows=None):
    """Literal oracle decode of the selected reference windows -> list of token lists."""
    wlen = omel.max_waveform_samples(o.encoder_ctx_size() - 10)
    ext = otr.window_extents(len(audio), 16000, wlen)
    rows = []
    for i, (s, e) in enumerate(ext):
        if windows is not None and i not in windows:
            continue
        t0 = time.time()
        mel = omel.prep_audio(torch.from_numpy(audio[s:e])[None], 16000.0)
        rows.append(otr.mels_to_tokens(o, pu.ost(st), mel, 10, beam, depth))
        print(f"   window {i}: {len(rows[-1]) - 4} tokens, {len(set(rows[-1][4:]))} distinct, {time.time() - t0:.1f} s",
              flush=True)
    return rows


def sec_wav(out):
    pcm16 = load_reference_wav()
    assert pcm16.shape[0] == 122276, pcm16.shape
    np.savez_compressed(os.path.join(HERE, "audio_16k_s16.npz"), pcm=pcm16)
    audio = pcm16.astype(np.float32) / np.float32(32767.0)
    mel = omel.prep_audio(torch.from_numpy(audio)[None])[0].numpy()          # [80, 764]
    out["mel_shape"] = np.array(mel.shape)
    out["mel_head"] = mel[:, :160].copy()                                     # first 1.6 s, all rows
    out["mel_strided"] = mel[::4, ::9].copy()
    ot = OracleWhisper(synth.synth_preset("tiny.en"))
    stt = SpecialTokens.for_vocab(51864)
    out["tiny_en_wav_greedy"] = np.array(otr.waveform_to_tokens(ot, pu.ost(stt), audio, 16000, 1, 100), np.int32)
    out["tiny_en_wav_beam5"] = np.array(otr.waveform_to_tokens(ot, pu.ost(stt), audio, 16000, 5, 24), np.int32)
    # the same clip on the checkpoint without the <|endoftext|> ramp: runs to max_depth
    on = OracleWhisper(synth.synth_preset("tiny.en", eot_beta=0.0))
    out["tiny_en_wav_greedy_long"] = np.array(otr.waveform_to_tokens(on, pu.ost(stt), audio, 16000, 1, 100), np.int32)
    out["tiny_en_wav_beam5_long"] = np.array(otr.waveform_to_tokens(on, pu.ost(stt), audio, 16000, 5, 40), np.int32)


def sec_micro(out):
    pcm16 = np.load(os.path.join(HERE, "audio_16k_s16.npz"))["pcm"]
    audio = pcm16.astype(np.float32) / np.float32(32767.0)
    mel = omel.prep_audio(torch.from_numpy(audio)[None])[0].numpy()
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    o = OracleWhisper(synth.synth_weights(dims, seed=4242))
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
    out["micro_greedy"] = np.array(otr.waveform_to_tokens(o, pu.ost(st), audio, 16000, 1, 24), np.int32)
    out["micro_beam5"] = np.array(otr.waveform_to_tokens(o, pu.ost(st), audio, 16000, 5, 24), np.int32)


def sec_workload(name):
    def run(out):
        wl = workloads.WORKLOADS[name]
        print(f"[{name}] {wl}", flush=True)
        o = OracleWhisper(wl.weights(), frame_limit_x2=wl.frame_limit_x2)
        st = SpecialTokens.for_vocab(o.dims.n_vocab)
        audio = synth.synth_audio(wl.n_samples, wl.audio_seed)
        rows = decode_windows(o, st, audio, wl.beam, wl.depth, wl.windows)
        toks, lens = pack(rows)
        out[f"{name}_tokens"] = toks
        out[f"{name}_lens"] = lens
        if wl.windows is None:
            stitched = []
            for r in rows:
                stitched = otr.stitch(stitched, r)                 # transcribe.rs:56-63
            out[f"{name}_stitched"] = np.array(stitched, np.int32)
    return run


SECTIONS = {"wav": sec_wav, "micro": sec_micro}
for _n in workloads.WORKLOADS:
    SECTIONS[_n] = sec_workload(_n)


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    want = sys.argv[1:] or list(SECTIONS)
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for stale in ("tiny_en_greedy", "tiny_en_beam5"):
        out.pop(stale, None)
    for name in want:
        t0 = time.time()
        SECTIONS[name](out)
        print(f"section {name}: {time.time() - t0:.1f} s", flush=True)
        np.savez_compressed(OUT, **out)
    for k, a in out.items():
        print(k, a.shape, a.dtype)


if __name__ == "__main__":
    main()
