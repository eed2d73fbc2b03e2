#!/bin/bash
# Round 3: numerics after the K-blocked MLP GEMM (log-prob distance of small / large-v2 to the f64 twin), the budget test,
# base.en with and without the persistent kernel, large-v2 450 s.   bash profiles/collect_r03e.sh   (through gpurun)
set -u
R=$PWD
OUT=$R/gpurun_out/r03e
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_budget.py -q -s > "$OUT/pytest_budget.log" 2>&1; tail -3 "$OUT/pytest_budget.log"
timeout 600 python whisper-burn_amd/tools/diag_batch_logprob.py small 60 > "$OUT/diag_small.log" 2>&1
timeout 900 python whisper-burn_amd/tools/diag_batch_logprob.py large-v2 16 > "$OUT/diag_large_v2.log" 2>&1
grep -h "vs f64\|total" "$OUT"/diag_*.log
for P in 0 1; do
  WHISPER_HIP_PERSIST=$P timeout 300 python bench.py --model base.en --steps 30 --warmup 3 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_base_en_persist$P.json"
done
timeout 900 python bench.py --model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_450s.json"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03e/bench_*.json")):
    try:
        j = json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["stages"]["encoder_ms_per_step"], j["stages"].get("encoder_frac_of_mfma_peak"))
    except Exception as e:
        print(f, "unreadable", e)
PY
