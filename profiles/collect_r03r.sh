#!/bin/bash
# Round 3: the split-precision GEMM bounded to two waves per SIMD (246 registers, two blocks of the 128 x 128 tile per CU)
# against the unbounded build (326 registers, one block).  Same arithmetic, same results.
set -u
R=$PWD
OUT=$R/gpurun_out/r03r
mkdir -p "$OUT"
for L in base w2; do
  LIB=$R/whisper-burn_amd/lib/libwhisper_hip.so; [ $L = w2 ] && LIB=$R/whisper-burn_amd/lib/libwhisper_hip_exp_w2.so
  WHISPER_HIP_LIB=$LIB WHISPER_HIP_ENCODER_SPLIT=1 timeout 300 python bench.py --model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_split_$L.json"
  WHISPER_HIP_LIB=$LIB WHISPER_HIP_ENCODER_SPLIT=1 timeout 300 python bench.py --model small --seconds 600 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_small_split_$L.json"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03r/bench_*.json")):
    try:
        j = json.load(open(f)); s = j["stages"]
        print(f, j["value"], j["ms_per_step"], "enc", round(s["encoder_ms_per_step"], 2), "ckv", round(s["cross_kv_ms_per_step"], 2), "enc TF", s.get("encoder_TFLOPs_algorithmic"))
    except Exception as e:
        print(f, "unreadable", e)
PY
