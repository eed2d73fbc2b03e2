#!/bin/bash
# Round 6: beam-path check after the 16-row MFMA MLP kernel: decode tests, the default bench's beam-5 leg with its kernel table,
# and config #3 (base.en, beam 5).
set -u
python -m pytest tests/test_gpu_workloads.py tests/test_gpu_edge.py tests/test_gpu_session.py tests/test_gpu_switches.py -x -q -m gpu > gpurun_out/r06_q_pytest.log 2>&1
tail -4 gpurun_out/r06_q_pytest.log
python bench.py --no-cpu-baseline --large-v2-leg off > gpurun_out/r06_q_bench.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_q_bench.json")); b = d["beam5"]
print(d["value"], d["ms_per_step"], "beam5", b["value"], b["ms_per_step"], b["config"]["tokens_checked"])
for k in b["config"]["kernels"]: print("  ", k["kernel"][:60], k["avg_launch_us"], k["launches_timed"])
PY
python bench.py --model base.en --beam 5 --max-depth 32 --no-cpu-baseline --large-v2-leg off --beam5-leg off --steps 20 > gpurun_out/r06_q_bench_base_beam5.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/r06_q_bench_base_beam5.json')); print('base.en beam5 depth32', d['value'], d['ms_per_step'])"
WHISPER_HIP_MLP16_MFMA=0 python bench.py --no-cpu-baseline --large-v2-leg off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MLP16_MFMA=0: beam5', d['beam5']['value'], d['beam5']['ms_per_step'])"
