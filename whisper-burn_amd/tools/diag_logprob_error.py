"""Developer diagnostic: where does the HIP path's log-prob error sit relative to the f32 oracle's own rounding?
134 KV-cached positions at tiny.en's real shape (NO_EOT checkpoint), one beam per window, random picks among the
top-5; prints max |hip - oracle_f32|, max |hip - oracle_f64|, max |oracle_f32 - oracle_f64|.
Run from the repo root on a GPU box: python whisper-burn_amd/tools/diag_logprob_error.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd"), os.path.join(ROOT, "tests")]
import parity_util as pu       # noqa: E402
import workloads               # noqa: E402
import whisper_burn_amd as wb  # noqa: E402
from oracle.model import OracleWhisper  # noqa: E402

wl = workloads.WORKLOADS["tiny_beam5"]
w = wl.weights()
eng = wb.Whisper.from_tensors(w)
o32, o64 = OracleWhisper(w), OracleWhisper(w, dtype=torch.float64)
st = wb.SpecialTokens.for_vocab(51864)
audio = wl.audio()
starts, lens = wb.window_extents(len(audio), 16000, 238559)
use = [0, 2]
sess = wb.Session.begin(eng, audio, starts[use], lens[use], max_beams=5)
sess.set_special_mask(st.is_special)
mels = pu.window_mels(o32, audio)
enc32 = [o32.forward_encoder(mels[i])[0] for i in use]
enc64 = [o64.forward_encoder(mels[i].double())[0] for i in use]
prompt = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
seqs = [[prompt[0]], [prompt[0]]]
rows = [[], []]
rng = np.random.default_rng(5)
for step in range(134):
    feeding = step < 3
    ids, lps = sess.step([s[-1] for s in seqs], [-1, -1] if step == 0 else [0, 1], [0, 1],
                         apply_special_mask=(not feeding) and step + 1 <= 5, k=0 if feeding else 5)
    for b in range(2):
        if feeding:
            seqs[b].append(prompt[step + 1])
        else:
            rows[b].append(sess.last_logprobs(b).copy())
            seqs[b].append(int(ids[b][int(rng.integers(0, 5))]))
sess.close()
for b in range(2):
    seq = seqs[b][:-1]
    r32 = pu.teacher_forced_logprobs(o32, st, enc32[b], seq)
    toks = torch.tensor([seq], dtype=torch.long)
    lg = o64.forward_decoder(toks, enc64[b][None])[0]
    maskv = torch.tensor(np.where(np.asarray(st.is_special).astype(bool), -np.inf, 0.0), dtype=torch.float64)
    r64 = np.stack([torch.log_softmax(lg[p] + (maskv if p + 1 <= 5 else 0.0), 0).numpy() for p in range(3, len(seq))])
    got = np.stack(rows[b])
    fin = np.isfinite(r64)
    d32 = np.abs(got - r32)[fin]; d64 = np.abs(got - r64)[fin]; dd = np.abs(r32 - r64)[fin]
    print(f"window {b}: rows {got.shape[0]}  max|lp| {np.abs(r64[fin]).max():.1f}  hip-f32 {d32.max():.3e}  hip-f64 {d64.max():.3e}  "
          f"f32-f64 {dd.max():.3e}   (99.99 pct: {np.quantile(d32, 0.9999):.3e} {np.quantile(d64, 0.9999):.3e} {np.quantile(dd, 0.9999):.3e})")
