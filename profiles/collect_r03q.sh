#!/bin/bash
# Round 3: tile choice of the split-precision GEMM (results are bit-identical across tiles: same K order per element),
# then the last sanity of the round: smoke() and the default bench line.
set -u
R=$PWD
OUT=$R/gpurun_out/r03q
mkdir -p "$OUT"
B="--model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8"
for T in 128 64; do
  WHISPER_HIP_ENCODER_SPLIT=1 WHISPER_HIP_SPLIT_TILE=$T timeout 300 python bench.py $B 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_split_tile$T.json"
done
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_tiny_en_30s.json"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03q/bench_*.json")):
    try:
        j = json.load(open(f)); s = j["stages"]
        print(f, j["value"], j["ms_per_step"], "enc", round(s["encoder_ms_per_step"], 2), "ckv", round(s["cross_kv_ms_per_step"], 2), "enc TF", s.get("encoder_TFLOPs_algorithmic"))
    except Exception as e:
        print(f, "unreadable", e)
PY
