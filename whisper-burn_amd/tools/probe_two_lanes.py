#!/usr/bin/env python3
"""Developer probe (GPU): does decoding a large-v2 window batch as TWO concurrent half-batches (two sessions, two streams, two host
threads -- ctypes drops the GIL) overlap one half's HBM-bound cross-K/V stream with the other half's latency-bound weight GEMMs?

    python whisper-burn_amd/tools/probe_two_lanes.py [seconds=450] [lanes=2]
"""
import os
import sys
import threading

os.environ.setdefault("WHISPER_HIP_GRAPH", "0")      # eager launches on both arms: two threads capturing graphs at once is another experiment
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd")]
import whisper_burn_amd as wb          # noqa: E402
from whisper_burn_amd import synth     # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 450.0
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = sys.argv[3] if len(sys.argv) > 3 else "large-v2"
w = synth.synth_preset(model, eot_beta=0.0)
eng = wb.Whisper.from_tensors(w)
del w
st = wb.SpecialTokens.for_vocab(eng.dims["n_vocab"])
params = wb.decode_params(st, beam_size=1, max_depth=100)
audio = synth.synth_audio(int(seconds * 16000), synth.BENCH_AUDIO_SEED + 5)
pcm = torch.from_numpy(audio).cuda()
n = len(audio)
wlen = wb.max_waveform_samples(1490)
starts, lens = wb.window_extents(n, 16000, wlen)
K = len(starts)


def run(lo, hi, out, i):
    out[i] = wb.waveform_to_tokens(eng, st, None, 16000, params=params, win_begin=lo, win_end=hi, device_ptr=pcm.data_ptr(),
                                   n_samples=n)[1]


def step(nl):
    cuts = [round(K * j / nl) for j in range(nl + 1)]
    out = [None] * nl
    th = [threading.Thread(target=run, args=(cuts[j], cuts[j + 1], out, j)) for j in range(nl)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return [r for part in out for r in part]


ref = step(1)
for nl in (1, lanes, 1, lanes):
    for _ in range(2):
        rows = step(nl)
    assert rows == ref, "lanes changed the tokens"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        step(nl)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{model} {seconds:g} s, {K} windows, lanes {nl}: {dt * 1e3:.1f} ms per step = {seconds / dt:.1f}x", flush=True)
