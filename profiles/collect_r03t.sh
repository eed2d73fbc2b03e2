#!/bin/bash
# Round 3: large-v2 450 s is 38 live rows = THREE row tiles of the skinny GEMM, so its figures before the plane-store fix
# (r03_h .. r03_m) had 6 early-ending windows too.  The A/B again, on one box, after the fix.
set -u
R=$PWD
OUT=$R/gpurun_out/r03t
mkdir -p "$OUT"
for S in 1 0; do
  WHISPER_HIP_BATCH_SKINNY=$S timeout 200 python bench.py --model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_450s_skinny$S.json"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03t/bench_*.json")):
    j = json.load(open(f)); s = j["stages"]
    print(f, j["value"], j["ms_per_step"], "enc", round(s["encoder_ms_per_step"], 1), "dec", round(s["decode_ms_per_step"], 1), j["kernels"][0]["kernel"][:40], j["kernels"][0]["avg_launch_us"])
PY
