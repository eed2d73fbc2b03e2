#!/bin/bash
# Round 3: the batch-mode layer cycle kernel by kernel (large-v2, 450 s), from a kernel trace.
set -u
R=$PWD
OUT=$R/gpurun_out/${RTAG:-r03l}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/p_kt -o kt -- python "$R/bench.py" --model large-v2 --seconds 450 --steps 1 --warmup 1 --no-cpu-baseline --mel-windows 8 > "$OUT/trace.log" 2>&1
DB=$(find /tmp/p_kt -name '*.db' | head -1)
python "$R/profiles/batch_layer_cycle.py" "$DB" > "$OUT/layer_cycle_large_v2.txt" 2>&1
python "$R/profiles/summarize_rocprof.py" "$DB" "$OUT/kernel_stats_large_v2_450s.csv"
cat "$OUT/layer_cycle_large_v2.txt"
cd "$R"
timeout 600 python bench.py --model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_450s.json"
timeout 600 python bench.py --model small --seconds 600 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_small_600s.json"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    j = json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["stages"]["decode_ms_per_step"])
PY
