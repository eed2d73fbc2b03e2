#!/bin/bash
# Round 4, sixth GPU call: is the slow large-v2 leg of r04e (encoder 593 ms instead of 198) a box phase or the range guard?
# Three separate processes of the large-v2 bench with clocks / power in between; then the log-prob budget tests that changed.
#   gpurun --timeout 700 -- 'bash profiles/collect_r04f.sh'
set -u
R=$PWD
OUT=$R/gpurun_out/r04f
mkdir -p "$OUT"
T0=$(date +%s)
smi() { /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | head -8; }
smi > $OUT/smi_0.txt
for i in 1 2 3; do
  timeout 150 python bench.py --model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --large-v2-leg off --beam5-leg off > $OUT/large_$i.json 2> $OUT/large_$i.err
  smi > $OUT/smi_$i.txt
  python - "$OUT/large_$i.json" <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print("large-v2:", o["value"], o["ms_per_step"], "encoder", o["stages"]["encoder_ms_per_step"], "decode", o["stages"]["decode_ms_per_step"],
          o["config"]["encoder_gemm"][:16], "->", o["config"]["encoder_gemm_after_the_run"])
except Exception as e:
    print("parse failed", e)
PY
done
cat $OUT/smi_0.txt $OUT/smi_3.txt
echo "[$(( $(date +%s) - T0 )) s] large-v2 runs done"
cd /tmp && export TMPDIR=/tmp
timeout 420 python -m pytest $R/tests/test_gpu_batchmode.py -m gpu -v -rA --durations=6 -p no:cacheprovider -k "logprob_rows or second_self_attention" > $OUT/pytest_rows.log 2>&1
grep -E "passed|failed|FAILED|PASSED|large-v2 |small " $OUT/pytest_rows.log | cut -c1-420 | tail -14
echo "[$(( $(date +%s) - T0 )) s] tests done"
