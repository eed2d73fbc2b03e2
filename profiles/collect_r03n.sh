#!/bin/bash
# Round 3, final build: the whole -m gpu suite, smoke(), the bench line (with its CPU baseline), the large-v2 / small legs.
#   bash profiles/collect_r03n.sh   (through gpurun, from the repo root)
set -u
R=$PWD
OUT=$R/gpurun_out/r03n
mkdir -p "$OUT"
python -m pytest tests -m gpu -q -s --durations=12 > "$OUT/pytest_gpu.log" 2>&1
tail -4 "$OUT/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
python bench.py > "$OUT/bench.log" 2>&1
grep '^{"metric' "$OUT/bench.log" > "$OUT/bench_tiny_en_30s.json"
python bench.py --model large-v2 --seconds 450 --steps 3 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_450s.json"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03n/bench_*.json")):
    j = json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"], j.get("cpu_baseline", {}).get("value") if j.get("cpu_baseline") else None)
PY
