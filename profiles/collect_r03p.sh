#!/bin/bash
# Round 3: clean A/B of the split-precision encoder on a fresh box + the `small` line after the plane-store fix.
set -u
R=$PWD
OUT=$R/gpurun_out/r03p
mkdir -p "$OUT"
B="--steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8"
for S in 0 1; do
  WHISPER_HIP_ENCODER_SPLIT=$S timeout 400 python bench.py --model large-v2 --seconds 450 $B 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_450s_split$S.json"
done
for S in 0 1; do
  WHISPER_HIP_ENCODER_SPLIT=$S timeout 300 python bench.py --model small --seconds 600 $B 2>&1 | grep '^{"metric' > "$OUT/bench_small_600s_split$S.json"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03p/bench_*.json")):
    try:
        j = json.load(open(f)); s = j["stages"]
        print(f, j["value"], j["ms_per_step"], "enc", round(s["encoder_ms_per_step"], 2), "ckv", round(s["cross_kv_ms_per_step"], 2), "dec", round(s["decode_ms_per_step"], 1), "enc TF", s.get("encoder_TFLOPs_algorithmic"))
    except Exception as e:
        print(f, "unreadable", e)
PY
