#!/bin/bash
# Round 5, the verification call: the whole -m gpu suite (as the driver runs it: -x -q) and the default bench line, on the
# commit they are run from.   gpurun --timeout 1500 -- 'bash profiles/collect_r05h.sh <tag>'
set -u
TAG=${1:-r05h}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
T0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
timeout 1150 python -m pytest $R/tests -x -q -m gpu -rA --durations=15 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -8
grep -E "^(small|large-v2|base.en)[ :]" $OUT/pytest_gpu.log | tail -12
echo "[$(( $(date +%s) - T0 )) s] suite done"
timeout 120 python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print("bench:", o["value"], "x,", o["ms_per_step"], "ms/step, steps", o["steps"], "; from host PCM:", o["from_host_pcm"]["value"])
    print("roofline:", {k: o["roofline"][k] for k in ("kernel", "frac", "avg_launch_us", "traffic", "algorithmic_bytes_per_launch")})
    print("cpu_baseline:", o["cpu_baseline"]["value"], o["cpu_baseline"]["cores"])
    b = o["beam5"]; print("beam5:", b["value"], b["ms_per_step"], b["config"]["generated_tokens_per_window"], b["config"]["stages_profiled_pass"]["launches"])
    l = o["large_v2"]; print("large_v2:", l["value"], l["ms_per_step"], "warm", l["warmup_step_ms"], "timed", l["step_ms"])
    print("   ", l["stages"]); print("   ", {k: l["roofline"][k] for k in ("kernel", "frac", "avg_launch_us", "traffic", "algorithmic_bytes_per_launch")})
    print("mel:", o["mel_frontend"]["value"], o["mel_frontend"]["frac_of_hbm_peak"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] bench done"
