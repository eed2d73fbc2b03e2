#!/usr/bin/env python3
"""Developer probe (GPU): a victim thread runs Session.begin (mel + encoder) repeatedly and compares its traced log-mel with the
quiet one, while an aggressor thread keeps the GPU busy with ONE kind of work on the same model.
WHISPER_HIP_ENC_TRACE=<dir> python whisper-burn_amd/tools/probe_threads_mel.py <enc|mel|begin|none> [rounds=200]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd")]
import whisper_burn_amd as wb          # noqa: E402
from whisper_burn_amd import synth     # noqa: E402
from whisper_burn_amd.model import max_waveform_samples, prep_audio   # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "enc"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
TRACE = os.environ["WHISPER_HIP_ENC_TRACE"]
dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
eng = wb.Whisper.from_tensors(synth.synth_weights(dims, seed=4242))
clip = synth.synth_audio(16000 * 47, 502)
win = max_waveform_samples(eng.max_mel_frames() - 12)
starts, lens = wb.window_extents(len(clip), 16000, win)


def read_mel():
    with open(os.path.join(TRACE, f"enc_trace_{threading.get_native_id()}.bin"), "rb") as f:
        while True:
            nm = f.read(32)
            if len(nm) < 32:
                return None
            nb = int(np.frombuffer(f.read(8), dtype=np.int64)[0])
            data = f.read(nb)
            if nm.split(b"\0")[0] == b"mel":
                return np.frombuffer(data, dtype=np.uint32).copy()


def begin():
    s = wb.Session.begin(eng, clip, starts, lens, 1, 12)
    m = read_mel()
    s.close()
    return m


ref = begin()
print("quiet repeatable:", np.array_equal(begin(), ref), flush=True)
mel_in = np.zeros((1, 80, eng.max_mel_frames()), dtype=np.float32)
mel_in[:, :, :] = np.linspace(-1, 1, mel_in.shape[2], dtype=np.float32)[None, None, :]
stop = threading.Event()
count = [0]


def aggressor():
    while not stop.is_set():
        if kind == "enc":
            eng.forward_encoder(mel_in)
        elif kind == "mel":
            prep_audio(clip[:win])
        elif kind == "begin":
            wb.Session.begin(eng, clip, starts, lens, 1, 12).close()
        else:
            stop.wait(0.01)
        count[0] += 1


bad = [0]


def victim():
    for rnd in range(rounds):
        m = begin()
        if not np.array_equal(m, ref):
            bad[0] += 1
            if bad[0] <= 6:
                idx = np.nonzero(m != ref)[0]
                fr = np.unique(idx % 1500 + 1500 * (idx // (80 * 1500)))    # (window, frame) keys
                print(f"round {rnd}: {len(idx)} words differ, frames (window*1500 + frame) {fr[:24].tolist()}", flush=True)
                mf, rf = m.view(np.float32).reshape(-1, 80, 1500), ref.view(np.float32).reshape(-1, 80, 1500)
                for key in fr[:4].tolist():
                    w, f = key // 1500, key % 1500
                    d = mf[w, :, f] - rf[w, :, f]
                    rows = np.nonzero(d)[0]
                    print(f"   window {w} frame {f} (pair {f % 32 // 2} of its tile): {len(rows)} mel rows differ: "
                          + " ".join(f"{r}:{d[r]:+.3f}" for r in rows[:80].tolist()), flush=True)


ta, tv = threading.Thread(target=aggressor), threading.Thread(target=victim)
ta.start(); tv.start(); tv.join(); stop.set(); ta.join()
print(f"aggressor {kind} ({count[0]} calls): victim log-mel differed in {bad[0]} of {rounds} rounds", flush=True)
