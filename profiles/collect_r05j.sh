#!/bin/bash
# Round 5, GPU call 10: K12 at large shapes -- 128 x 128 tiles with 64-deep k-tiles on eight waves (one block per CU) against
# the default (32-deep, four waves, two blocks per CU).
set -u
REPS=2 bash profiles/ab.sh r05j_k12bk64 "--model large-v2 --seconds 450 --max-depth 20 --steps 3 --warmup 2 --beam5-leg off" WHISPER_HIP_SPLIT_TILE=0 WHISPER_HIP_SPLIT_TILE=12864
REPS=1 bash profiles/ab.sh r05j_k12bk64_small "--model small --seconds 600 --steps 3 --warmup 1 --beam5-leg off" WHISPER_HIP_SPLIT_TILE=0 WHISPER_HIP_SPLIT_TILE=12864
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_r05j_k12bk64*/variant*_rep*.log")):
    try:
        o = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
        st = o["stages"]
        print(f.split("/")[-2][8:], f.split("/")[-1], o["value"], o["ms_per_step"], "enc", st["encoder_ms_per_step"], "ckv", st["cross_kv_ms_per_step"], "frac", st["encoder_frac_of_mfma_peak"])
    except Exception as e:
        print(f, "failed", e)
PY
