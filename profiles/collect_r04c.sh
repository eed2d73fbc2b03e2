#!/bin/bash
# Round 4, third GPU call: A/B of the round's decode changes against the r04b build on ONE box (per-wave LayerNorm + dead-row
# prefetch skip in the persistent kernel; 20 / 16 loads in flight in the skinny GEMM + one-round-trip resolve-LN), the tests
# added since r04b, the default bench line.   gpurun --timeout 1500 -- 'bash profiles/collect_r04c.sh'
set -u
R=$PWD
OUT=$R/gpurun_out/r04c
mkdir -p "$OUT"
T0=$(date +%s)
REPS=3 bash profiles/ab.sh r04c_tiny "--steps 100 --warmup 5 --large-v2-leg off --beam5-leg off" whisper-burn_amd/lib/libwhisper_hip_base.so whisper-burn_amd/lib/libwhisper_hip.so
REPS=2 bash profiles/ab.sh r04c_large "--model large-v2 --seconds 450 --steps 2 --warmup 1" whisper-burn_amd/lib/libwhisper_hip_base.so whisper-burn_amd/lib/libwhisper_hip.so
REPS=1 bash profiles/ab.sh r04c_small "--model small --seconds 600 --steps 3 --warmup 1" whisper-burn_amd/lib/libwhisper_hip_base.so whisper-burn_amd/lib/libwhisper_hip.so
echo "[$(( $(date +%s) - T0 )) s] A/B done"
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_gpu_handoff.py $R/tests/test_gpu_shard_rccl.py $R/tests/test_gpu_switches.py $R/tests/test_gpu_batchmode.py $R/tests/test_gpu_parity.py -m gpu -q --durations=8 -p no:cacheprovider > $OUT/pytest_new.log 2>&1
tail -22 $OUT/pytest_new.log
echo "[$(( $(date +%s) - T0 )) s] tests done"
cd $R
( time timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    o = json.loads([l for l in open("gpurun_out/r04c/bench_default.json") if l.startswith('{"metric"')][-1])
    print("bench:", o["value"], o["ms_per_step"], "roofline", o["roofline"]["frac"], o["roofline"]["avg_launch_us"])
    print("beam5:", o["beam5"]["value"], o["beam5"]["ms_per_step"], o["beam5"]["config"]["generated_tokens_per_window"])
    lv = o["large_v2"]
    print("large_v2:", lv["value"], lv["ms_per_step"], lv["roofline"]["kernel"], lv["roofline"]["frac"], lv["stages"]["decode_ms_per_step_untraced"], lv["stages"]["decode_frac_of_hbm_peak"])
    for k in lv["kernels"][:5]: print("   ", k["kernel"], k["avg_launch_us"], k["frac_of_hbm_peak"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] bench done"
