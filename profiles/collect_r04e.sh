#!/bin/bash
# Round 4, last GPU call: the whole -m gpu suite and the default bench line on the round's final build.
#   gpurun --timeout 1500 -- 'bash profiles/collect_r04e.sh'
set -u
R=$PWD
OUT=$R/gpurun_out/r04e
mkdir -p "$OUT"
T0=$(date +%s)
( time timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    o = json.loads([l for l in open("gpurun_out/r04e/bench_default.json") if l.startswith('{"metric"')][-1])
    print("bench:", o["value"], o["ms_per_step"], "roofline", o["roofline"]["frac"], o["roofline"]["avg_launch_us"], "traffic", o["roofline"]["traffic"])
    print("beam5:", o["beam5"]["value"], "large_v2:", o["large_v2"]["value"], o["large_v2"]["roofline"]["frac"], "cpu:", o["cpu_baseline"]["value"])
    print("mel:", o["mel_frontend"]["value"], "stages:", o["stages"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] bench done"
cd /tmp && export TMPDIR=/tmp
timeout 1150 python -m pytest $R/tests -m gpu -v -rA --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -15
echo "[$(( $(date +%s) - T0 )) s] suite done"
cd $R && python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
