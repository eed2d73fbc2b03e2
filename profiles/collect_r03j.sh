#!/bin/bash
# Round 3: batch mode with the skinny GEMM (+ GELU prologue) and the streaming cross-attention blocks that fold / normalise /
# project their own query.   bash profiles/collect_r03j.sh   (through gpurun)
set -u
R=$PWD
OUT=$R/gpurun_out/r03j
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_switches.py -q -k "batch_mode" > "$OUT/pytest_quick.log" 2>&1; tail -3 "$OUT/pytest_quick.log"
timeout 600 python -m pytest tests/test_gpu_batchmode.py -q -s -k "stream_9x1" >> "$OUT/pytest_quick.log" 2>&1; tail -3 "$OUT/pytest_quick.log"
for F in 1 0; do
  WHISPER_HIP_CROSS_STREAM_FUSE=$F timeout 600 python bench.py --model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_450s_fuse$F.json"
  WHISPER_HIP_CROSS_STREAM_FUSE=$F timeout 600 python bench.py --model small --seconds 600 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_small_600s_fuse$F.json"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03j/bench_*.json")):
    try:
        j = json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["stages"]["decode_ms_per_step"], j["stages"].get("decode_kernels_per_token"))
        for k in j.get("kernels", [])[:7]: print("   ", k["kernel"], k["share_of_decode_kernel_time"], k["avg_launch_us"], k["frac_of_hbm_peak"])
    except Exception as e:
        print(f, "unreadable", e)
PY
