#!/bin/bash
# Round 5, GPU call 8: K12 at large shapes -- 128 x 64 tiles with three / four k-tiles in flight against the 128 x 128 default.
set -u
R=$PWD
REPS=2 bash profiles/ab.sh r05i_k12tile "--model large-v2 --seconds 450 --max-depth 20 --steps 3 --warmup 2 --beam5-leg off" WHISPER_HIP_SPLIT_TILE=0 WHISPER_HIP_SPLIT_TILE=12864 WHISPER_HIP_SPLIT_TILE=12865
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_r05i_k12tile/variant*_rep*.log")):
    try:
        o = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
        st = o["stages"]
        print(f.split("/")[-1], o["value"], o["ms_per_step"], "enc", st["encoder_ms_per_step"], "ckv", st["cross_kv_ms_per_step"], "frac", st["encoder_frac_of_mfma_peak"])
    except Exception as e:
        print(f, "failed", e)
PY
