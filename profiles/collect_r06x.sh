#!/bin/bash
# Round 6, the final tree (per-device GPU turn): the same commands as collect_r06t.sh -- the whole -m gpu suite, smoke(), the default bench and the
# driver's invocation.
set -u
python -m pytest tests -x -q -m gpu > gpurun_out/r06_x_pytest_gpu.log 2>&1
tail -32 gpurun_out/r06_x_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_x_smoke.log 2>&1; tail -1 gpurun_out/r06_x_smoke.log
python bench.py > gpurun_out/r06_x_bench_default.json 2> gpurun_out/r06_x_bench_default.err; tail -2 gpurun_out/r06_x_bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_x_bench_steps20.json 2>/dev/null
python - <<'PY'
import json
for f in ("r06_x_bench_default", "r06_x_bench_steps20"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["config"]["tokens_checked"], "roofline", d["roofline"]["frac"], "beam5", d["beam5"]["value"],
          d["beam5"]["config"]["tokens_checked"], "large", d["large_v2"]["value"], d["large_v2"]["config"]["tokens_checked"],
          "mel", d["mel_frontend"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
