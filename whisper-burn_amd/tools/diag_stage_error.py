"""Developer diagnostic: which stage carries the HIP path's distance from the (f64-evaluated) reference algorithm?
tiny.en real shape, first window of the beam workload.  Each stage is fed the SAME input on both sides.
Run from the repo root on a GPU box: python whisper-burn_amd/tools/diag_stage_error.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd"), os.path.join(ROOT, "tests")]
import workloads               # noqa: E402
import whisper_burn_amd as wb  # noqa: E402
from oracle import mel as omel  # noqa: E402
from oracle.model import OracleWhisper  # noqa: E402

wl = workloads.WORKLOADS["tiny_beam5"]
w = wl.weights()
eng = wb.Whisper.from_tensors(w)
o32, o64 = OracleWhisper(w), OracleWhisper(w, dtype=torch.float64)
audio = wl.audio()
starts, lens = wb.window_extents(len(audio), 16000, 238559)
x = np.asarray(audio[starts[0]:starts[0] + lens[0]], np.float32)


def mx(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


# ---- mel ----
m_hip = wb.prep_audio(x, 16000.0)
m_o32 = omel.prep_audio(torch.from_numpy(x)[None], 16000.0).numpy()
m_f64 = omel.prep_audio_f64(x, 16000.0)
m_hip = np.asarray(m_hip).reshape(m_o32.shape)
print(f"mel      : hip-f64 {mx(m_hip, m_f64):.3e}   o32-f64 {mx(m_o32, m_f64):.3e}   hip-o32 {mx(m_hip, m_o32):.3e}")


def clip_pad(m):
    m = np.asarray(m)
    m = (m[None] if m.ndim == 2 else m)[:, :, :1490]
    return np.concatenate([m, np.zeros((1, 80, 10), m.dtype)], 2)


# ---- encoder on the same (o32) mel ----
mel_in = clip_pad(m_o32).astype(np.float32)
e_hip = eng.forward_encoder(mel_in)
e_o32 = o32.forward_encoder(torch.from_numpy(mel_in)).numpy()
e_f64 = o64.forward_encoder(torch.from_numpy(mel_in).double()).numpy()
print(f"encoder  : hip-f64 {mx(e_hip, e_f64):.3e}   o32-f64 {mx(e_o32, e_f64):.3e}   max|enc| {np.abs(e_f64).max():.2f}")
# ---- encoder error caused by the mel difference alone (f64 model on both) ----
e_f64_hipmel = o64.forward_encoder(torch.from_numpy(clip_pad(m_hip).astype(np.float64))).numpy()
e_f64_f64mel = o64.forward_encoder(torch.from_numpy(clip_pad(m_f64).astype(np.float64))).numpy()
print(f"enc(mel) : f64 model, hip mel vs f64 mel {mx(e_f64_hipmel, e_f64_f64mel):.3e}   o32 mel vs f64 mel {mx(e_f64, e_f64_f64mel):.3e}")

# ---- decoder on the same (o32) encoder output ----
g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_outputs.npz"))
toks = g["tiny_beam5_tokens"][0][:int(g["tiny_beam5_lens"][0])][:104].astype(np.int64)[None]
d_hip = eng.forward_decoder(toks.astype(np.int32), e_o32)
d_o32 = o32.forward_decoder(torch.from_numpy(toks), torch.from_numpy(e_o32)).numpy()
d_f64 = o64.forward_decoder(torch.from_numpy(toks), torch.from_numpy(e_o32).double()).numpy()
print(f"decoder  : hip-f64 {mx(d_hip, d_f64):.3e}   o32-f64 {mx(d_o32, d_f64):.3e}   max|logit| {np.abs(d_f64).max():.1f}  (stateless, {toks.shape[1]} tokens)")
# ---- logits error caused by the encoder difference alone (f64 decoder on both) ----
d_f64_hipenc = o64.forward_decoder(torch.from_numpy(toks), torch.from_numpy(e_hip).double()).numpy()
print(f"dec(enc) : f64 decoder, hip enc vs f64 enc {mx(d_f64_hipenc, o64.forward_decoder(torch.from_numpy(toks), torch.from_numpy(e_f64)).numpy()):.3e}")
d_f64_hipmel = o64.forward_decoder(torch.from_numpy(toks), torch.from_numpy(e_f64_hipmel)).numpy()
d_f64_f64mel = o64.forward_decoder(torch.from_numpy(toks), torch.from_numpy(e_f64_f64mel)).numpy()
d_f64_o32mel = o64.forward_decoder(torch.from_numpy(toks), torch.from_numpy(e_f64)).numpy()
print(f"dec(mel) : f64 model, logits from hip mel vs f64 mel {mx(d_f64_hipmel, d_f64_f64mel):.3e}   from o32 mel vs f64 mel {mx(d_f64_o32mel, d_f64_f64mel):.3e}   hip mel vs o32 mel {mx(d_f64_hipmel, d_f64_o32mel):.3e}")
