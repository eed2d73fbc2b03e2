#!/usr/bin/env python3
"""Developer tool: random-geometry parity fuzzing of the HIP sources on a GPU-less machine.

Runs the engine through the hipemu functional model (tools/hipemu; build it with `make -C whisper-burn_amd/tools/hipemu`)
on randomly drawn model shapes (d in {128, 384, 512, 768}: every decode-kernel template family), vocabulary sizes that
are not multiples of any tile, context sizes, clip lengths, beam widths, depths and switches, and compares with the
oracle: token ids of waveform_to_tokens, and -- when they differ -- the oracle's own top-2 gap at the first differing
step, so that a near-tie (gap < 1e-3: the synthetic checkpoints of random shapes are not tuned for margins the way the
golden workloads are) is told apart from a defect.  Test infrastructure only: never imported by the product.

    python whisper-burn_amd/tools/emu_fuzz.py --seed 1 --minutes 30 [--guard] [--reverse]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "whisper-burn_amd")

CHILD = r"""
import json, sys
import numpy as np, torch
sys.path[:0] = [%(root)r, %(pkg)r, %(tests)r]
import parity_util as pu
import whisper_burn_amd as wb
from oracle import transcribe as otr
from oracle.model import OracleWhisper
from whisper_burn_amd import synth
cfg = json.loads(sys.argv[1])
dims = synth.micro_dims(n_state=cfg["d"], n_head=cfg["d"] // 64, n_layer=cfg["layers"], n_vocab=cfg["vocab"],
                        n_audio_ctx=cfg["audio_ctx"], n_text_ctx=cfg["text_ctx"])
w = synth.synth_weights(dims, seed=cfg["wseed"])
eng = wb.Whisper.from_tensors(w)
o = OracleWhisper(w, ln_eps_inside_sqrt=bool(cfg["ln_inside"]), frame_limit_x2=bool(cfg["x2"]))
eng.set_layernorm_variant(bool(cfg["ln_inside"]))
eng.set_frame_limit(bool(cfg["x2"]))
st = wb.SpecialTokens.for_vocab(cfg["vocab"])
audio = synth.synth_audio(cfg["samples"], cfg["aseed"])
res = {"status": "ok"}
try:
    ref, rw = otr.waveform_to_tokens(o, pu.ost(st), audio, 16000, cfg["beam"], cfg["depth"], return_windows=True)
    ref_err = None
except AssertionError as e:
    ref, rw, ref_err = None, None, "assert"
try:
    got, wins = wb.waveform_to_tokens(eng, st, audio, 16000, cfg["beam"], cfg["depth"])
    got_err = None
except wb.WbError as e:
    got, wins, got_err = None, None, e.status
if (ref_err is None) != (got_err is None):
    res = {"status": "error-mismatch", "oracle": ref_err, "engine": got_err}
elif ref_err is None and wins != rw:
    # first differing window / position; the oracle's top-2 gap there (greedy only: the teacher-forced row)
    wi = next(i for i, (a, b) in enumerate(zip(wins, rw)) if a != b) if len(wins) == len(rw) else -1
    gap = None
    if wi >= 0 and cfg["beam"] == 1:
        a, b = wins[wi], rw[wi]
        p = next((j for j, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
        mels = pu.window_mels(o, audio)
        enc = o.forward_encoder(mels[wi])[0]
        if 4 <= p < len(b):
            lp = pu.teacher_forced_logprobs(o, st, enc, b[:p + 1])
            row = lp[p - 4]
            top2 = np.sort(row[np.isfinite(row)])[-2:]
            gap = float(top2[1] - top2[0])
    res = {"status": "tie" if (gap is not None and gap < 1e-3) else "TOKEN-MISMATCH", "window": wi, "gap": gap,
           "engine": wins[wi] if wi >= 0 else [len(x) for x in wins], "oracle": rw[wi] if wi >= 0 else [len(x) for x in rw]}
elif ref_err is None and got != ref:
    res = {"status": "STITCH-MISMATCH"}
# stateless forward on a random prefix (logits <= 1e-3)
if res["status"] == "ok" and cfg["samples"] >= 400:
    n = min(cfg["samples"], 16000 * 3)
    mel = wb.prep_audio(audio[None, :n])
    L = min(cfg["text_ctx"], 3 + cfg["depth"] %% 7 + 1)
    rng = np.random.default_rng(cfg["aseed"])
    toks = rng.integers(0, cfg["vocab"] - 16, size=(1, L)).astype(np.int32)
    mel = mel[:, :, :eng.max_mel_frames()]
    lg = eng.forward(mel, toks)
    rl = o.forward(torch.from_numpy(mel), torch.from_numpy(toks)).numpy()
    err = float(np.abs(lg - rl).max())
    if not err < 1e-3:
        res = {"status": "LOGIT-MISMATCH", "err": err}
eng.close()
print("RESULT " + json.dumps(res))
"""


def draw_many(rng):
    """Batch mode: 9-16 short windows (n_audio_ctx = 400: 3.9 s windows, 0.9 s apart), greedy or beams -- more than 8 live rows
    (up to 64: one to four 16-row tiles of the skinny GEMM), i.e. the skinny / tiled split-K MFMA GEMMs, resolve-LN, the
    streaming (one beam; with or without its fused front) or chunked (beams) cross-attention, dec_topk_rows."""
    d = int(rng.choice([128, 128, 384, 512, 768]))
    n_win = int(rng.integers(9, 17)) if d == 128 else int(rng.integers(9, 12))
    secs = 3.9 + 0.9 * (n_win - 1) - float(rng.random()) * 0.8
    return dict(d=d, layers=int(rng.integers(1, 3)) if d == 128 else 1, vocab=int(rng.choice([515, 1031, 2053])),
                audio_ctx=400, text_ctx=int(rng.choice([16, 64, 448])), x2=0, ln_inside=int(rng.random() < 0.3),
                samples=int(secs * 16000), aseed=int(rng.integers(0, 1 << 30)), wseed=int(rng.integers(0, 1 << 30)),
                beam=int(rng.choice([1, 1, 1, 2, 3, 4])), depth=int(rng.integers(1, 11)),
                switches=str(rng.choice(["", "", "", "WHISPER_HIP_CROSS_STREAM=0", "WHISPER_HIP_CHAIN=0", "WHISPER_HIP_GRAPH=0",
                                         "WHISPER_HIP_BATCH_SKINNY=0", "WHISPER_HIP_CROSS_STREAM_FUSE=0",
                                         "WHISPER_HIP_ENCODER_SPLIT=1", "WHISPER_HIP_SK_PAIR=1"])))


def draw(rng):
    d = int(rng.choice([128, 128, 384, 512, 768]))
    audio_ctx = int(rng.choice([400, 1500])) if d == 128 else 400     # (emulated MFMA: keep the encoder affordable)
    x2 = int(rng.random() < 0.3)
    win_s = (audio_ctx * (2 if x2 else 1) - 10) / 100.0
    secs = float(rng.choice([0.02, 0.03, 0.5, win_s * 0.5, win_s, win_s + 0.4, win_s * 2.2, win_s * 3.1]))
    if d >= 512:
        secs = min(secs, 6.0)
    return dict(d=d, layers=int(rng.integers(1, 3)) if d <= 384 else 1, vocab=int(rng.choice([515, 1031, 2053, 4099])),
                audio_ctx=audio_ctx, text_ctx=int(rng.choice([16, 64, 448])), x2=x2, ln_inside=int(rng.random() < 0.3),
                samples=int(secs * 16000) + int(rng.integers(0, 160)), aseed=int(rng.integers(0, 1 << 30)),
                wseed=int(rng.integers(0, 1 << 30)), beam=int(rng.choice([1, 1, 2, 3, 5])), depth=int(rng.integers(0, 13)),
                switches=str(rng.choice(["", "", "", "WHISPER_HIP_FUSE_SUB=0", "WHISPER_HIP_FUSE_X=0", "WHISPER_HIP_CHAIN=0",
                                         "WHISPER_HIP_GRAPH=0", "WHISPER_HIP_CROSS_STREAM=0", "WHISPER_HIP_FUSE_Q=0",
                                         "WHISPER_HIP_PERSIST=0", "WHISPER_HIP_ENCODER_SPLIT=1"])))


def main():
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10)
    ap.add_argument("--guard", action="store_true")
    ap.add_argument("--reverse", action="store_true")
    ap.add_argument("--many-windows", action="store_true", help="batch-mode cases: 9-16 windows per clip")
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    lib = os.path.join(PKG, "lib", "libwhisper_hip_emu.so")
    code = CHILD % {"root": ROOT, "pkg": PKG, "tests": os.path.join(ROOT, "tests")}
    t_end = time.time() + args.minutes * 60
    counts = {}
    log = open(args.log, "a") if args.log else None
    n = 0
    while time.time() < t_end:
        cfg = draw_many(rng) if args.many_windows else draw(rng)
        env = {k: v for k, v in os.environ.items() if not k.startswith("WHISPER_HIP_")}
        env.update(WHISPER_HIP_LIB=lib, WHISPER_HIP_ALLOW_EMU="1")
        if cfg["switches"]:
            k, v = cfg["switches"].split("=")
            env[k] = v
        if args.guard:
            env.update(HIPEMU_GUARD="1", HIPEMU_SEGV_TRACE="1")
        if args.reverse:
            env["HIPEMU_ORDER"] = "reverse"
        t0 = time.time()
        try:
            p = subprocess.run([sys.executable, "-c", code, json.dumps(cfg)], env=env, capture_output=True, text=True,
                               timeout=900)
            lines = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            res = json.loads(lines[-1][7:]) if lines else {"status": "CRASH", "rc": p.returncode, "stderr": p.stderr[-1500:]}
        except subprocess.TimeoutExpired:
            res = {"status": "TIMEOUT"}
        n += 1
        counts[res["status"]] = counts.get(res["status"], 0) + 1
        line = json.dumps({"n": n, "t": round(time.time() - t0, 1), "cfg": cfg, "res": res})
        if res["status"] not in ("ok",):
            print(line, flush=True)
        if log:
            log.write(line + "\n")
            log.flush()
    print("SUMMARY", json.dumps(counts), flush=True)


if __name__ == "__main__":
    main()
