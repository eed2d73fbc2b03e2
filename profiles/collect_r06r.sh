python bench.py --no-cpu-baseline --large-v2-leg off > gpurun_out/r06_r_bench.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_r_bench.json")); b = d["beam5"]
print(d["value"], d["ms_per_step"], "beam5", b["value"], b["ms_per_step"], b["config"]["tokens_checked"])
for k in b["config"]["kernels"]: print("  ", k["kernel"][:60], k["avg_launch_us"], k["launches_timed"])
PY
