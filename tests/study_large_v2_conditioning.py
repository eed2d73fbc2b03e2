#!/usr/bin/env python3
"""CPU study (not a test; run by hand, result recorded in LABLOG R6.1): conditioning of the large-v2 synthetic checkpoints.

For each variant -- the round-5 fixture (logit scale depth-normalised to 2.6) and the round-6 variant `logit_depth_norm=False`
(logit scale held at 6, only branch gains / attention strengths depth-normalised) -- the f32 oracle against the f64 evaluation of
the same operators over a top-5 random walk at full window length, and the magnitude of the log-probs being compared.

    python tests/study_large_v2_conditioning.py [n_tokens]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd"), os.path.join(ROOT, "tests")]

from oracle import mel as omel                       # noqa: E402
from oracle.model import OracleWhisper, log_softmax  # noqa: E402
from whisper_burn_amd import synth                   # noqa: E402
from whisper_burn_amd.tokens import SpecialTokens    # noqa: E402


def walk(w, L, seed=5):
    o32, o64 = OracleWhisper(w), OracleWhisper(w, dtype=torch.float64)
    st = SpecialTokens.for_vocab(o32.dims.n_vocab)
    audio = synth.synth_audio(160 * 1490 + 100, 1240)
    mel = omel.prep_audio(torch.from_numpy(audio[:160 * 1490])[None], 16000.0)
    mel = torch.cat([mel, torch.zeros(1, 80, 10)], 2)
    xa32, xa64 = o32.forward_encoder(mel), o64.forward_encoder(mel)
    enc_err = float((xa32.double() - xa64).abs().max())
    maskv = torch.tensor(np.where(np.asarray(st.is_special).astype(bool), -np.inf, 0.0))
    seq = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
    rng = np.random.default_rng(seed)
    while len(seq) < L:
        lg = o32.forward_decoder(torch.tensor([seq]), xa32)[0, -1].double()
        if len(seq) <= 5:
            lg = lg + maskv
        seq.append(int(torch.topk(lg, 5).indices[int(rng.integers(0, 5))]))
    toks = torch.tensor([seq])
    l32, l64 = o32.forward_decoder(toks, xa32)[0], o64.forward_decoder(toks, xa64)[0]
    worst, mag, gap = 0.0, 0.0, np.inf
    for p in range(3, L):
        m = maskv if p + 1 <= 5 else 0.0
        a, b = log_softmax(l32[p].double() + m, 0), log_softmax(l64[p] + m, 0)
        fin = torch.isfinite(b)
        worst = max(worst, float((a[fin] - b[fin]).abs().max()))
        mag = max(mag, float(b[fin].abs().max()))
        t2 = torch.topk(b[fin], 2).values
        gap = min(gap, float(t2[0] - t2[1]))
    return {"enc_err": enc_err, "f32_vs_f64_worst": worst, "max_abs_logprob": mag, "min_top2_gap": gap,
            "distinct": len(set(seq[4:])), "n": L - 4}


if __name__ == "__main__":
    L = 4 + (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
    torch.set_num_threads(os.cpu_count() or 1)
    for name, kw in (("round-5 fixture (logit scale 2.6)", {}), ("logit_depth_norm=False (logit scale 6)", {"logit_depth_norm": False})):
        t0 = time.time()
        r = walk(synth.synth_preset("large-v2", eot_beta=0.0, **kw), L)
        print(name, r, f"{time.time() - t0:.0f} s", flush=True)
