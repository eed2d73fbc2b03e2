#!/bin/bash
# Round 4, second GPU call: the new default build (split-precision encoder incl. conv2, two-level P.V sums):
# stage-split numerics, the default bench line (with the large-v2 leg and the depth-100 CPU run), the whole -m gpu suite.
#   gpurun --timeout 1700 -- 'bash profiles/collect_r04b.sh'
set -u
R=$PWD
OUT=$R/gpurun_out/r04b
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
D=$R/whisper-burn_amd/tools/diag_stage_split.py
T0=$(date +%s)
for model in large-v2 small; do
  c=/tmp/diag_$model.npz
  timeout 500 python $D $model $c base 36 > $OUT/diag_${model}_base.log 2>&1
  WHISPER_HIP_ENCODER_SPLIT=0 timeout 300 python $D $model $c f32enc 36 > $OUT/diag_${model}_f32enc.log 2>&1
  grep -h "SUMMARY" -A 8 $OUT/diag_${model}_*.log | grep -v "^--"
  grep -h "w[04] enc:" $OUT/diag_${model}_*.log
done
echo "[$(( $(date +%s) - T0 )) s] diag done"
cd $R
( time timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
tail -c 600 $OUT/bench_default.err
python - <<'PY'
import json
try:
    o = json.loads([l for l in open("gpurun_out/r04b/bench_default.json") if l.startswith('{"metric"')][-1])
    print("bench:", o["value"], o["ms_per_step"], "roofline", o["roofline"]["frac"], "cpu", o["cpu_baseline"]["value"], o["cpu_baseline"]["depth32"]["value"])
    lv = o["large_v2"]
    print("large_v2:", lv["value"], lv["ms_per_step"], lv["roofline"]["kernel"], lv["roofline"]["frac"], lv["stages"])
    print("mel:", o["mel_frontend"]["value"], "stages", o["stages"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] bench done"
timeout 900 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
echo "[$(( $(date +%s) - T0 )) s] suite done"
