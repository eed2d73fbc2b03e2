#!/usr/bin/env python3
"""Developer probe (GPU): two host threads decode different clips with ONE model at the same time; prints where a threaded result
differs from the single-threaded one.   python whisper-burn_amd/tools/probe_threads.py [beam=1] [rounds=30]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd")]
import whisper_burn_amd as wb          # noqa: E402
from whisper_burn_amd import synth     # noqa: E402

beam = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
eng = wb.Whisper.from_tensors(synth.synth_weights(dims, seed=4242))
st = wb.SpecialTokens.for_vocab(1031)
clips = [synth.synth_audio(16000 * 35, 501), synth.synth_audio(16000 * 47, 502)]
ref = [wb.waveform_to_tokens(eng, st, c, 16000, beam, 12) for c in clips]
again = [wb.waveform_to_tokens(eng, st, c, 16000, beam, 12) for c in clips]
print("single-threaded repeatable:", again == ref, flush=True)
bad = 0
for rnd in range(rounds):
    out = [None, None]

    def run(i):
        try:
            out[i] = wb.waveform_to_tokens(eng, st, clips[i], 16000, beam, 12)
        except Exception as e:      # noqa: BLE001
            out[i] = repr(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        if out[i] != ref[i]:
            bad += 1
            if isinstance(out[i], str):
                print(f"round {rnd} clip {i}: ERROR {out[i][:300]}", flush=True)
                continue
            gw, rw = out[i][1], ref[i][1]
            for w, (g, r) in enumerate(zip(gw, rw)):
                if g != r:
                    k = next((j for j, (a, b) in enumerate(zip(g, r)) if a != b), min(len(g), len(r)))
                    print(f"round {rnd} clip {i} window {w}/{len(rw)}: first difference at position {k}: got {g[k:k+4]} ref {r[k:k+4]} (len {len(g)} vs {len(r)})", flush=True)
print(f"beam {beam}: {bad} mismatching results in {rounds} rounds x 2 threads")
