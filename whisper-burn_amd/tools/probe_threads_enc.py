#!/usr/bin/env python3
"""Developer probe (GPU): two host threads run Session.begin (mel + encoder + cross K/V) on different clips with ONE model at the
same time; compares every window's encoder output bit for bit with the single-threaded one.
python whisper-burn_amd/tools/probe_threads_enc.py [rounds=100]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd")]
import whisper_burn_amd as wb          # noqa: E402
from whisper_burn_amd import synth     # noqa: E402
from whisper_burn_amd.model import max_waveform_samples   # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
eng = wb.Whisper.from_tensors(synth.synth_weights(dims, seed=4242))
engs = [eng, wb.Whisper.from_tensors(synth.synth_weights(dims, seed=4242)) if os.environ.get("PROBE_TWO_MODELS") else eng]
clips = [synth.synth_audio(16000 * 35, 501), synth.synth_audio(16000 * 47, 502)]
if os.environ.get("PROBE_SAME_CLIP"):
    clips[0] = clips[1]
if os.environ.get("PROBE_SAME_LEN"):
    clips[0] = synth.synth_audio(16000 * 47, 501)
win = max_waveform_samples(eng.max_mel_frames() - 12)


TRACE = os.environ.get("WHISPER_HIP_ENC_TRACE")
MODE = os.environ.get("PROBE_MODE", "fresh")     # fresh | persistent | solo | nod2h
WIDTH = {"conv1": 128, "qkv": 384, "mlp1": 512}


def read_trace():
    """Stages of this thread's last encoder pass (engine.cpp: trace_flush)."""
    out = []
    with open(os.path.join(TRACE, f"enc_trace_{threading.get_native_id()}.bin"), "rb") as f:
        while True:
            nm = f.read(32)
            if len(nm) < 32:
                break
            nb = int(np.frombuffer(f.read(8), dtype=np.int64)[0])
            out.append((nm.split(b"\0")[0].decode(), np.frombuffer(f.read(nb), dtype=np.uint32).copy()))
    return out


def report_trace(tag, got, ref):
    for (name, g), (_, r) in zip(got, ref):
        if g.shape == r.shape and np.array_equal(g, r):
            continue
        if name in ("mel", "mel0"):
            idx = np.nonzero(g != r)[0]
            fr = np.unique(idx % 1500)
            PAIR_HIST.update(((fr % 32) // 2).tolist())
            print(f"  {tag}: {name} differs in {len(idx)} words; frames {fr[:40].tolist()}; pairs-in-tile histogram so far "
                  f"{sorted(PAIR_HIST.items())}", flush=True)
            return
        w = WIDTH.get(name.split(".")[-1], 128)
        idx = np.nonzero(g != r)[0]
        rows, cols = idx // w, idx % w
        print(f"  {tag}: first differing stage {name}: {len(idx)} of {len(g)} words, rows {rows.min()}..{rows.max()} "
              f"({len(np.unique(rows))} rows; 64-row tiles {sorted(set((rows // 64).tolist()))[:12]}), cols {cols.min()}..{cols.max()} "
              f"(64-col tiles {sorted(set((cols // 64).tolist()))})", flush=True)
        gf, rf = g.view(np.float32), r.view(np.float32)
        k = idx[:6]
        print(f"     first words at {k.tolist()}: got {gf[k].tolist()} ref {rf[k].tolist()}", flush=True)
        return
    print(f"  {tag}: every traced stage identical", flush=True)


traces = {}
import collections   # noqa: E402
PAIR_HIST = collections.Counter()


def enc(i):
    starts, lens = wb.window_extents(len(clips[i]), 16000, win)
    s = wb.Session.begin(engs[i], clips[i], starts, lens, 1, 12)
    if TRACE:
        traces[i] = read_trace()
    if MODE == "nod2h":
        s.close()
        return []
    out = [s.encoder_output(w).copy() for w in range(len(starts))]
    s.close()
    return out


ref = [enc(i) for i in range(2)]
ref_traces = dict(traces)
again = [enc(i) for i in range(2)]
print("single-threaded repeatable:", all(np.array_equal(a, b) for x, y in zip(ref, again) for a, b in zip(x, y)), flush=True)
bad = 0
per = [0, 0]
for rnd in range(rounds):
    out = [None, None]

    def run(i):
        try:
            out[i] = enc(i)
        except Exception as e:      # noqa: BLE001
            out[i] = repr(e)

    if MODE == "persistent":
        if rnd == 0:
            import queue
            qs = [queue.Queue() for _ in range(2)]
            done = queue.Queue()

            def worker(i):
                while qs[i].get():
                    run(i)
                    done.put(i)

            workers = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(2)]
            [t.start() for t in workers]
        [q.put(True) for q in qs]
        [done.get() for _ in range(2)]
    else:
        th = [threading.Thread(target=run, args=(i,)) for i in (range(2) if MODE != "solo" else [1])]
        [t.start() for t in th]
        [t.join() for t in th]
        if MODE == "solo":
            out[0] = ref[0]
    for i in range(2):
        if isinstance(out[i], str):
            bad += 1
            print(f"round {rnd} clip {i}: ERROR {out[i][:300]}", flush=True)
            continue
        if MODE == "nod2h":
            tr, rt = dict(traces[i]), dict(ref_traces[i])
            if not np.array_equal(tr["mel"], rt["mel"]):
                bad += 1
                per[i] += 1
            continue
        if TRACE and any(not np.array_equal(g, r) for g, r in zip(out[i], ref[i])):
            report_trace(f"round {rnd} clip {i}", traces[i], ref_traces[i])
        for w, (g, r) in enumerate(zip(out[i], ref[i])):
            if not np.array_equal(g, r):
                bad += 1
                per[i] += 1
                diff = np.argwhere(g != r)
                rows = np.unique(diff[:, 0]); cols = np.unique(diff[:, 1])
                if bad > 12:
                    continue
                print(f"round {rnd} clip {i} window {w}: {len(diff)} elements differ, rows {rows.min()}..{rows.max()} "
                      f"({len(rows)}), cols {cols.min()}..{cols.max()} ({len(cols)}), max |d| "
                      f"{np.abs(g - r).max():.3g} nan {int(np.isnan(g).sum())}", flush=True)
print("per clip:", per, flush=True)
print(f"encoder: {bad} mismatching windows in {rounds} rounds x 2 threads")
