#!/bin/bash
# Round 3: skinny GEMM, modes apart: 0 = tiled GEMM + fold launches, 2 = skinny GEMM (planes) + fold launches,
# 1 = skinny GEMM with last-arriver epilogues.   bash profiles/collect_r03i.sh   (through gpurun)
set -u
R=$PWD
OUT=$R/gpurun_out/r03i
mkdir -p "$OUT"
for S in 2 1; do
  WHISPER_HIP_BATCH_SKINNY=$S timeout 600 python bench.py --model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_450s_skinny$S.json"
done
WHISPER_HIP_BATCH_SKINNY=2 timeout 600 python bench.py --model small --seconds 600 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_small_600s_skinny2.json"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03i/bench_*.json")):
    try:
        j = json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["stages"]["decode_ms_per_step"], j["stages"].get("decode_kernels_per_token"))
        for k in j.get("kernels", [])[:7]: print("   ", k["kernel"], k["share_of_decode_kernel_time"], k["avg_launch_us"], k["frac_of_hbm_peak"])
    except Exception as e:
        print(f, "unreadable", e)
PY
