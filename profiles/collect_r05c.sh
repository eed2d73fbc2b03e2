#!/bin/bash
# Round 5, GPU call 3: K12 register prefetch (PF = 4) for the small-tile configurations + dec_topk_rows batched loads, against
# the library of the previous commit (lib/libwhisper_hip_base.so); quick parity; the default bench with the settled warm-up.
set -u
R=$PWD; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
L=whisper-burn_amd/lib
T0=$(date +%s)
REPS=3 bash profiles/ab.sh r05c_pf "--steps 40 --warmup 3 --large-v2-leg off" $L/libwhisper_hip_base.so $L/libwhisper_hip.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_r05c_pf/variant*_rep*.log")):
    try:
        o = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
        b = o["beam5"]
        tk = [k["avg_launch_us"] for k in b["config"]["kernels"] if "topk_rows" in k["kernel"]]
        print(f.split("/")[-1], o["value"], "enc ms", o["stages"]["encoder_ms_per_step"], "ckv", o["stages"]["cross_kv_ms_per_step"], "beam5", b["value"], "topk_rows us", tk)
    except Exception as e:
        print(f, "failed", e)
PY
REPS=2 bash profiles/ab.sh r05c_pf_base "--model base.en --steps 20 --warmup 3 --large-v2-leg off --beam5-leg off" $L/libwhisper_hip_base.so $L/libwhisper_hip.so
echo "[$(( $(date +%s) - T0 )) s] A/B done"
( cd /tmp && export TMPDIR=/tmp
timeout 500 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_golden.py $R/tests/test_gpu_workloads.py $R/tests/test_gpu_session.py $R/tests/test_gpu_scale.py -m gpu -x -q -p no:cacheprovider \
  -k "not large and not small_10min" 2>&1 | tail -6 ) | tee $OUT/pytest_quick.log
echo "[$(( $(date +%s) - T0 )) s] quick tests done"
( cd /tmp && export TMPDIR=/tmp && timeout 500 python $R/bench.py --steps 40 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err )
python - $OUT/bench_default.json <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print("bench:", o["value"], "x,", o["ms_per_step"], "ms/step; stages", o["stages"]["encoder_ms_per_step"], o["stages"]["decode_ms_per_step"])
    b = o["beam5"]; print("beam5:", b["value"], b["ms_per_step"], b["config"]["generated_tokens_per_window"])
    l = o["large_v2"]; print("large_v2:", l["value"], l["ms_per_step"], "warm", l["warmup_step_ms"], "timed", l["step_ms"], "enc", l["stages"]["encoder_ms_per_step"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] bench done"
