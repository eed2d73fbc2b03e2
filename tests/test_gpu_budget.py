"""GPU: the PCM -> log-prob error budget, from RAW AUDIO on both sides (no shared log-mel).

The model-parity tests at 1e-3 (test_gpu_workloads.py, test_gpu_batchmode.py) hand the HIP log-mel to both sides,
because the oracle's own frontend -- the reference's dense f32 DFT (audio.rs:349-364) -- is the largest single
rounding term of the whole path and the synthetic checkpoints amplify a log-mel difference ~25x into the logits
(DESIGN.md section 5).  This file backs that shortcut with assertions instead of prose.  Three evaluations of the
SAME algorithm on the SAME samples, teacher-forced over the HIP path's own greedy row:

  hip   the product path from PCM (HIP mel -> encoder -> KV-cached session steps), f32
  o32   the oracle from PCM (oracle f32 mel -> f32 encoder -> stateless f32 decoder)
  o64   the exact twin: the same operators evaluated in f64 from the f64 log-mel (oracle.mel.prep_audio_f64)

Asserted, at tiny.en's real shape, on the bench audio's first window AND on the reference's own audio.wav:
  |hip - o64| <= 1e-3              the HIP path is within the north star's tolerance of the EXACT result from raw audio
  |hip - o64| <  |o32 - o64|       ... and strictly closer to it than the reference-style f32 evaluation is
  |hip - o32| <= |hip - o64| + |o32 - o64|             the two f32 paths differ by no more than their own roundings
                                                       (on the bench clip that sum is < BUDGET = 3e-3; on the reference's
                                                       audio.wav, whose upper mel bands sit at the clamp floor, the f32 oracle
                                                       alone is ~1e-2 from the exact result -- measured 1.06e-2 -- while the
                                                       HIP path is 1.5e-4 from it)
"""
import numpy as np
import pytest
import torch

import whisper_burn_amd as wb
import parity_log
import workloads
from oracle import mel as omel
from oracle.model import OracleWhisper
from test_oracle_golden import golden

pytestmark = pytest.mark.gpu

EXACT_TOL = 1e-3        # north_star: logits within 1e-3 (fp32), here against the exact (f64) evaluation, from PCM
BUDGET = 3e-3           # hip vs oracle-f32, both from PCM: bounded by the sum of the two distances to the exact result
WLEN = 238559


def _rows(o, st, mel, row, dtype):
    """Teacher-forced masked log-softmax rows (transcribe.rs:271-284) of `row` in `dtype` from a [1, 80, T] log-mel."""
    mel = torch.as_tensor(mel).to(dtype)
    keep = min(mel.shape[2], o.encoder_ctx_size() - 10)
    melp = torch.cat([mel[:, :, :keep], torch.zeros(1, 80, 10, dtype=dtype)], 2)        # transcribe.rs:171-177
    enc = o.forward_encoder(melp)
    lg = o.forward_decoder(torch.tensor([row], dtype=torch.long), enc)[0]
    maskv = torch.tensor(np.where(np.asarray(st.is_special).astype(bool), -np.inf, 0.0), dtype=dtype)
    out = []
    for p in range(3, len(row) - 1):
        v = lg[p] + (maskv if p + 1 <= 5 else 0.0)
        out.append(torch.log_softmax(v.double(), 0).numpy() if dtype == torch.float64 else
                   (v - v.max() - torch.log(torch.exp(v - v.max()).sum())).numpy())
    return np.stack(out)


@pytest.mark.parametrize("clip", ["bench_window0", "reference_audio_wav"])
def test_pcm_to_logprob_budget_tiny_en(clip):
    w = workloads.WORKLOADS["tiny_bench"].weights()
    eng = wb.Whisper.from_tensors(w)
    o32, o64 = OracleWhisper(w), OracleWhisper(w, dtype=torch.float64)
    st = wb.SpecialTokens.for_vocab(51864)
    audio = workloads.WORKLOADS["tiny_bench"].audio()[:WLEN] if clip == "bench_window0" else golden()[1]
    audio = np.ascontiguousarray(audio, np.float32)
    assert len(audio) <= WLEN
    _, wins = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 48)
    row = wins[0]
    assert len(row) >= 6          # (the reference clip ends on <|endoftext|> after a few tokens with this checkpoint)
    # hip: every step's full log-prob row from the KV-cached session, PCM in
    starts, lens = wb.window_extents(len(audio), 16000, WLEN)
    sess = wb.Session.begin(eng, audio, starts[:1], lens[:1], max_beams=1)
    sess.set_special_mask(st.is_special)
    hip = []
    for p in range(len(row) - 1):
        sess.step([row[p]], [-1 if p == 0 else 0], [0], apply_special_mask=(p >= 3 and p + 1 <= 5), k=1 if p >= 3 else 0)
        if p >= 3:
            hip.append(sess.last_logprobs(0).copy())
    sess.close(); eng.close()
    hip = np.stack(hip)
    r32 = _rows(o32, st, omel.prep_audio(torch.from_numpy(audio)[None]), row, torch.float32)
    r64 = _rows(o64, st, omel.prep_audio_f64(audio)[None], row, torch.float64)
    fin = np.isfinite(r64)
    assert (np.isfinite(hip) == fin).all() and (np.isfinite(r32) == fin).all()
    with np.errstate(invalid="ignore"):          # (-inf - -inf at the masked special tokens: excluded by `fin`)
        d_hip_exact = float(np.abs(hip - r64)[fin].max())
        d_o32_exact = float(np.abs(r32 - r64)[fin].max())
        d_hip_o32 = float(np.abs(hip - r32)[fin].max())
    parity_log.record(f"budget::pcm_to_logprob_tiny_en[{clip}] hip vs exact (f64, from PCM)", d_hip_exact, EXACT_TOL,
                      float(np.abs(r64[fin]).max()), n_rows=hip.shape[0], oracle_f32_vs_exact=d_o32_exact, hip_vs_oracle_f32=d_hip_o32)
    print(f"{clip}: rows {hip.shape[0]}, max |log-prob| {np.abs(r64[fin]).max():.1f}; hip-exact {d_hip_exact:.3e}, "
          f"oracle_f32-exact {d_o32_exact:.3e}, hip-oracle_f32 {d_hip_o32:.3e}")
    assert d_hip_exact <= EXACT_TOL, d_hip_exact
    assert d_hip_exact < d_o32_exact, (d_hip_exact, d_o32_exact)
    assert d_hip_o32 <= d_hip_exact + d_o32_exact + 1e-6, (d_hip_o32, d_hip_exact, d_o32_exact)
    if clip == "bench_window0":
        assert d_hip_o32 <= BUDGET, d_hip_o32
    # decisions are unaffected: the row is the argmax chain under all three evaluations
    for name, r in (("o32", r32), ("o64", r64)):
        assert [int(np.argmax(x)) for x in r] == row[4:], name
