"""Developer diagnostic (GPU box): batch-mode session log-prob rows of `small` / large-v2 against the f32 oracle AND the
f64 evaluation of the same algorithm, by position -- is a difference above 1e-3 rounding (the f32 oracle is as far from
the exact result) or a defect?   python whisper-burn_amd/tools/diag_batch_logprob.py small|large-v2 [n_steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd"), os.path.join(ROOT, "tests")]
import parity_util as pu       # noqa: E402
import whisper_burn_amd as wb  # noqa: E402
from oracle.model import OracleWhisper  # noqa: E402
from whisper_burn_amd import synth  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "small"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
WLEN = 238559
t0 = time.time()
w = synth.synth_preset(model, eot_beta=0.0)
eng, o32 = wb.Whisper.from_tensors(w), OracleWhisper(w)
st = wb.SpecialTokens.for_vocab(eng.dims["n_vocab"])
audio = synth.synth_audio(1900000, 1240)
starts, lens = wb.window_extents(len(audio), 16000, WLEN)
print(f"{model}: setup {time.time() - t0:.0f} s", flush=True)
mels = pu.window_mels(o32, audio, frontend=wb.prep_audio)
prompt = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]


def run(use, max_beams, fork_at, seed, label, f64_rows=1):
    sess = wb.Session.begin(eng, audio, starts[use], lens[use], max_beams=max_beams)
    sess.set_special_mask(st.is_special)
    beams = [([prompt[0]], i) for i in range(len(use))]
    parents = [-1] * len(use)
    rng = np.random.default_rng(seed)
    rec = []
    for step in range(n_steps):
        feeding = step < 3
        ids, lps = sess.step([b[0][-1] for b in beams], parents, [b[1] for b in beams],
                             apply_special_mask=(not feeding) and step + 1 <= 5, k=0 if feeding else 5)
        if feeding:
            beams = [(b[0] + [prompt[step + 1]], b[1]) for b in beams]
            parents = list(range(len(beams)))
            continue
        for slot, (seq, wdx) in enumerate(beams):
            rec.append((tuple(seq), wdx, sess.last_logprobs(slot).copy()))
        nxt, npar = [], []
        for slot, (seq, wdx) in enumerate(beams):
            pick = int(rng.integers(0, 5))
            nxt.append((seq + [int(ids[slot][pick])], wdx)); npar.append(slot)
            if max_beams > 1 and step == fork_at:
                nxt.append((seq + [int(ids[slot][(pick + 1) % 5])], wdx)); npar.append(slot)
        order = sorted(range(len(nxt)), key=lambda i: nxt[i][1])
        beams, parents = [nxt[i] for i in order], [npar[i] for i in order]
    sess.close()
    finals = sorted({(r[0], r[1]) for r in rec}, key=lambda x: -len(x[0]))
    rows32 = {}
    encs = {}
    for seq, wdx in finals:
        if (seq, wdx) in rows32:
            continue
        if wdx not in encs:
            encs[wdx] = o32.forward_encoder(mels[use[wdx]])[0]
        lp = pu.teacher_forced_logprobs(o32, st, encs[wdx], list(seq))
        for n in range(4, len(seq) + 1):
            rows32.setdefault((seq[:n], wdx), lp[n - 4])
    by_pos = {}
    for seq, wdx, got in rec:
        ref = rows32[(seq, wdx)]
        fin = np.isfinite(ref)
        dabs = np.abs(got[fin] - ref[fin])
        # differences of the row's five best tokens relative to the best one: what decisions depend on (a uniform offset cancels)
        top = np.argsort(-ref)[:5]
        drel = np.abs((got[top] - got[top[0]]) - (ref[top] - ref[top[0]])).max()
        b = by_pos.setdefault(len(seq) // 16, [0.0, 0.0, 0.0])
        b[0] = max(b[0], float(dabs.max())); b[1] = max(b[1], float(drel)); b[2] = max(b[2], float(np.abs(ref[fin]).max()))
    print(f"[{label}] vs oracle f32, by position bucket (16): " +
          "  ".join(f"{16 * k}+: abs {v[0]:.2e} top5-rel {v[1]:.2e} max|lp| {v[2]:.0f}" for k, v in sorted(by_pos.items())), flush=True)
    # the exact twin on the longest f64_rows sequences
    o64 = OracleWhisper(w, dtype=torch.float64)
    maskv = torch.tensor(np.where(np.asarray(st.is_special).astype(bool), -np.inf, 0.0), dtype=torch.float64)
    for seq, wdx in finals[:f64_rows]:
        enc64 = o64.forward_encoder(mels[use[wdx]].double())[0]
        lg = o64.forward_decoder(torch.tensor([list(seq)], dtype=torch.long), enc64[None])[0]
        r64 = np.stack([torch.log_softmax(lg[p] + (maskv if p + 1 <= 5 else 0.0), 0).numpy() for p in range(3, len(seq))])
        worst = [0.0, 0.0, 0.0]
        for s2, w2, got in rec:
            if w2 == wdx and s2 == seq[:len(s2)]:
                ref64 = r64[len(s2) - 4]; ref32 = rows32[(s2, w2)]
                fin = np.isfinite(ref64)
                worst[0] = max(worst[0], float(np.abs(got[fin] - ref64[fin]).max()))
                worst[1] = max(worst[1], float(np.abs(ref32[fin] - ref64[fin]).max()))
                worst[2] = max(worst[2], float(np.abs(got[fin] - ref32[fin]).max()))
        print(f"[{label}] window {use[wdx]} vs f64: hip-exact {worst[0]:.3e}  oracle_f32-exact {worst[1]:.3e}  hip-oracle_f32 {worst[2]:.3e}", flush=True)
    del o64


run(list(range(9)), 1, -1, 21, "9 windows x 1 beam (streaming cross-attention)")
run([0, 2, 4, 6, 9], 2, 3, 11, "5 windows x 2 beams (chunked + combine)")
run([0, 2], 1, -1, 5, "2 windows x 1 beam (small-batch GEMV path)")
print(f"total {time.time() - t0:.0f} s")
