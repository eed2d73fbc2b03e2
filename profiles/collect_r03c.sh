#!/bin/bash
# Round 3: first GPU runs of the persistent flag-chained decode kernel -- parity at tiny.en's real shape, A/B against the
# launch-per-sublayer chain on the same box, role timeline.   bash profiles/collect_r03c.sh   (through gpurun)
set -u
R=$PWD
OUT=$R/gpurun_out/r03c
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_session.py tests/test_gpu_switches.py -q -x -k "tiny_bench or bench_workload or greedy or chain or switch" > "$OUT/pytest_persist.log" 2>&1
tail -3 "$OUT/pytest_persist.log"
for i in 1 2; do
  WHISPER_HIP_PERSIST=0 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_chain_$i.json"
  WHISPER_HIP_PERSIST=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_persist_$i.json"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c/bench_*.json")):
    try:
        j = json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["roofline"]["kernel"][:40], j["roofline"]["frac"], j["e2e_roofline"]["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
WHISPER_HIP_PERSIST=1 WHISPER_HIP_PS_STAMPS=$OUT/stamps.bin timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 > "$OUT/bench_stamps.log" 2>&1
python profiles/ps_timeline.py "$OUT/stamps.bin" > "$OUT/ps_timeline.txt" 2>&1
cat "$OUT/ps_timeline.txt"
