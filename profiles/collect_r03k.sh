#!/bin/bash
# Round 3: runtime knobs that touch the kernel-boundary cost (graph replay of 388 launches per large-v2 step).
set -u
R=$PWD
OUT=$R/gpurun_out/r03k
mkdir -p "$OUT"
B="--model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8"
timeout 600 python bench.py $B 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_base.json"
HIP_FORCE_DEV_KERNARG=1 timeout 600 python bench.py $B 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_devkernarg.json"
HSA_ENABLE_INTERRUPT=0 timeout 600 python bench.py $B 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_nointr.json"
WHISPER_HIP_GRAPH=0 timeout 600 python bench.py $B 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_nograph.json"
T="--steps 30 --warmup 5 --no-cpu-baseline --mel-windows 8"
timeout 300 python bench.py $T 2>&1 | grep '^{"metric' > "$OUT/bench_tiny_base.json"
HSA_ENABLE_INTERRUPT=0 timeout 300 python bench.py $T 2>&1 | grep '^{"metric' > "$OUT/bench_tiny_nointr.json"
timeout 300 python bench.py --model small --seconds 600 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_small_600s.json"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03k/bench_*.json")):
    try:
        j = json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["stages"]["decode_ms_per_step"], j["stages"].get("decode_kernels_per_token"))
    except Exception as e:
        print(f, "unreadable", e)
PY
