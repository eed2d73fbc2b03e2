"""Developer diagnostic (GPU box): where does the log-prob distance between the HIP path and the exact (f64) evaluation
come from at `small` / large-v2 -- the encoder or the decoder, the KV-cached batch-mode kernels or the stateless ones?

    python whisper-burn_amd/tools/diag_stage_split.py <model> <cache.npz> [label] [n_steps]

First call (cache absent): drives a batch-mode session (9 windows x 1 beam) with a random pick among each row's top 5,
then evaluates every row's sequence with the f64 twin and the f32 oracle ON THE HIP ENCODER OUTPUT (so the decoder is
the only difference) and on their own encoder outputs, and stores sequences + reference rows in the cache.  Later calls
(other switch settings, label them) force the same token sequences through the session and compare with the cache.

Per window it prints, by position, max over the vocabulary of
  enc       |enc_hip - enc64| (rms, max) next to |enc32 - enc64|
  dec_sess  |session row   - f64 decoder(enc_hip)|     the KV-cached decode kernels alone
  dec_st    |stateless hip - f64 decoder(enc_hip)|     wb_forward_decoder alone
  dec_o32   |f32 oracle decoder(enc_hip) - f64 decoder(enc_hip)|   what a CPU f32 decoder loses
  enc_eff   |f64 decoder(enc_hip) - f64 decoder(enc64)|  what the HIP encoder's rounding costs downstream
  enc_eff32 |f64 decoder(enc32)   - f64 decoder(enc64)|  what the oracle encoder's rounding costs downstream
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd"), os.path.join(ROOT, "tests")]
import parity_util as pu       # noqa: E402
import whisper_burn_amd as wb  # noqa: E402
from oracle.model import OracleWhisper, log_softmax  # noqa: E402
from whisper_burn_amd import synth  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "small"
cache = sys.argv[2] if len(sys.argv) > 2 else "/tmp/diag_stage_%s.npz" % model
label = sys.argv[3] if len(sys.argv) > 3 else "base"
n_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 36
NW = 9
t0 = time.time()
if model == "micro":        # functional-model dry run of this script (WHISPER_HIP_ALLOW_EMU=1)
    w = synth.synth_weights(synth.micro_dims(n_audio_ctx=400, n_text_ctx=64), seed=77)
else:
    w = synth.synth_preset(model, eot_beta=0.0)
eng = wb.Whisper.from_tensors(w)
st = wb.SpecialTokens.for_vocab(eng.dims["n_vocab"])
WLEN = wb.max_waveform_samples(eng.dims["n_audio_ctx"] - 10)      # 238559 at n_audio_ctx = 1500
audio = synth.synth_audio(1900000 if model != "micro" else 48000 + NW * (WLEN - 48000), 1240)
starts, lens = wb.window_extents(len(audio), 16000, WLEN)
assert len(starts) >= NW, len(starts)
use = list(range(NW))
prompt = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
maskv64 = torch.tensor(np.where(np.asarray(st.is_special).astype(bool), -np.inf, 0.0), dtype=torch.float64)
have = os.path.exists(cache)
print(f"[{label}] {model}: setup {time.time() - t0:.0f} s, cache {'hit' if have else 'miss'}", flush=True)
cz = dict(np.load(cache)) if have else {}


def rows_of(o, enc, seq, dt):
    lg = o.forward_decoder(torch.tensor([list(seq)], dtype=torch.long), torch.as_tensor(enc).to(dt)[None])[0].double()
    return np.stack([log_softmax(lg[p] + (maskv64 if p + 1 <= 5 else 0.0), 0).numpy() for p in range(3, len(seq))])


def hip_stateless_rows(enc, seq):
    lg = torch.from_numpy(eng.forward_decoder(np.asarray([seq], np.int32), enc[None])[0]).double()
    return np.stack([log_softmax(lg[p] + (maskv64 if p + 1 <= 5 else 0.0), 0).numpy() for p in range(3, len(seq))])


# ---- the session: 9 windows x 1 beam = batch mode with the streaming cross-attention kernel
sess = wb.Session.begin(eng, audio, starts[use], lens[use], max_beams=1)
sess.set_special_mask(st.is_special)
seqs = [[prompt[0]] for _ in use]
rng = np.random.default_rng(21)
rec = [[] for _ in use]
for step in range(n_steps):
    feeding = step < 3
    ids, lps = sess.step([s[-1] for s in seqs], [-1] * NW if step == 0 else list(range(NW)), list(range(NW)),
                         apply_special_mask=(not feeding) and step + 1 <= 5, k=0 if feeding else 5)
    for i in range(NW):
        if feeding:
            seqs[i].append(prompt[step + 1])
            continue
        rec[i].append(sess.last_logprobs(i).astype(np.float64))
        if have:
            seqs[i].append(int(cz[f"seq{i}"][len(seqs[i])]))
        else:
            seqs[i].append(int(ids[i][int(rng.integers(0, 5))]))
enc_hip = [sess.encoder_output(i) for i in range(NW)]
sess.close()
print(f"[{label}] session done {time.time() - t0:.0f} s", flush=True)

if not have:
    o32, o64 = OracleWhisper(w), OracleWhisper(w, dtype=torch.float64)
    mels = pu.window_mels(o32, audio, frontend=wb.prep_audio)
    for i in range(NW):
        seq = seqs[i][:-1]                   # rows exist for prefixes of length 4 .. len - 1
        cz[f"seq{i}"] = np.asarray(seqs[i], np.int64)
        enc32 = o32.forward_encoder(mels[use[i]])[0].numpy()
        enc64 = o64.forward_encoder(mels[use[i]].double())[0].numpy()
        cz[f"enc32_{i}"], cz[f"enc64_{i}"], cz[f"enc_base_{i}"] = enc32, enc64, enc_hip[i]
        cz[f"r64_base_{i}"] = rows_of(o64, enc_hip[i], seq, torch.float64)      # f64 decoder on the BASE run's hip encoder output
        cz[f"r32_base_{i}"] = rows_of(o32, enc_hip[i], seq, torch.float32)
        cz[f"r64_e64_{i}"] = rows_of(o64, enc64, seq, torch.float64)
        cz[f"r64_e32_{i}"] = rows_of(o64, enc32, seq, torch.float64)
        cz[f"r32_e32_{i}"] = rows_of(o32, enc32, seq, torch.float32)
        print(f"[{label}] oracle rows window {i} {time.time() - t0:.0f} s", flush=True)
    np.savez(cache, **cz)
    del o32, o64


def perpos(a, b):
    fin = np.isfinite(b) & np.isfinite(a)
    return np.where(fin, np.abs(np.where(fin, a, 0) - np.where(fin, b, 0)), 0).max(1)


def fmt(v):
    return " ".join(f"{x:.1e}" for x in v)


agg = {}
for i in range(NW):
    seq = [int(t) for t in cz[f"seq{i}"][:-1]]
    enc64, enc32 = cz[f"enc64_{i}"], cz[f"enc32_{i}"]
    got = np.stack(rec[i])
    eh, e3 = enc_hip[i].astype(np.float64) - enc64, enc32.astype(np.float64) - enc64
    print(f"[{label}] w{i} enc: hip rms {np.sqrt((eh ** 2).mean()):.2e} max {np.abs(eh).max():.2e} | "
          f"oracle_f32 rms {np.sqrt((e3 ** 2).mean()):.2e} max {np.abs(e3).max():.2e}", flush=True)
    same_enc = np.array_equal(enc_hip[i], cz[f"enc_base_{i}"])
    r64_e64 = cz[f"r64_e64_{i}"]
    if not same_enc:                     # another encoder arithmetic: its own exact decoder rows
        if "o64" not in globals():
            o64 = OracleWhisper(w, dtype=torch.float64)
        cz[f"r64_base_{i}"] = rows_of(o64, enc_hip[i], seq, torch.float64)
        same_enc = True
    lines = {"total_sess": perpos(got, r64_e64), "total_o32": perpos(cz[f"r32_e32_{i}"], r64_e64),
             "enc_eff32": perpos(cz[f"r64_e32_{i}"], r64_e64)}
    if same_enc:
        r64_h = cz[f"r64_base_{i}"]
        lines["dec_sess"] = perpos(got, r64_h)
        if np.array_equal(enc_hip[i], cz[f"enc_base_{i}"]):
            lines["dec_o32"] = perpos(cz[f"r32_base_{i}"], r64_h)
        lines["enc_eff"] = perpos(r64_h, r64_e64)
        if i in (0, 4) and label == "base":
            lines["dec_st"] = perpos(hip_stateless_rows(enc_hip[i], seq), r64_h)
    for k, v in lines.items():
        agg.setdefault(k, []).append(v)
        if i in (0, 4):
            print(f"[{label}] w{i} {k:10s} {fmt(v)}", flush=True)
print(f"[{label}] SUMMARY over {NW} windows x {n_steps - 3} positions (per-position max over the vocabulary):")
for k, vs in agg.items():
    a = np.concatenate(vs)
    print(f"[{label}]   {k:10s} max {a.max():.2e}  p90 {np.quantile(a, 0.9):.2e}  median {np.median(a):.2e}  rms {np.sqrt((a ** 2).mean()):.2e}  (n={len(a)})")
print(f"[{label}] total {time.time() - t0:.0f} s")
