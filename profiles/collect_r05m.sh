#!/bin/bash
# Round 5: the bench line exactly as the driver invokes it (N = 1), on the shipped library, plus smoke().
set -u
R=$PWD; OUT=$R/gpurun_out/r05m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 100 python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s)
timeout 500 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
echo "bench wall $(( $(date +%s) - T0 )) s"
python - $OUT/bench_steps20.json <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print(o["value"], o["ms_per_step"], o["steps"], o["roofline"]["frac"], "beam5", o["beam5"]["value"], "large", o["large_v2"]["value"], o["large_v2"]["warmup_step_ms"], "cpu", o["cpu_baseline"]["value"])
PY
