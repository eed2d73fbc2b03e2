#!/bin/bash
# Round 5, GPU call 2: the split-precision decoder GEMM (dec_skinny_f16x3_kernel, batch-mode decode on the 16-bit matrix path).
# 1. quick parity subset   2. A/B vs the exact-f32 skinny kernel (large-v2 450 s, small 600 s)   3. the default bench line
# with its new legs (host PCM, depth-100 beam 5 + its kernel table, large-v2)   4. K12 counters   5. the whole -m gpu suite
set -u
R=$PWD; OUT=$R/gpurun_out/r05b; mkdir -p $OUT
T0=$(date +%s)
( cd /tmp && export TMPDIR=/tmp
timeout 400 python -m pytest $R/tests/test_gpu_switches.py $R/tests/test_gpu_e2e.py $R/tests/test_gpu_session.py -m gpu -x -q -p no:cacheprovider \
  -k "DECODER_SPLIT or default or own_frontend or many_windows" 2>&1 | tail -15 ) | tee $OUT/pytest_quick.log
echo "[$(( $(date +%s) - T0 )) s] quick subset done"
REPS=2 bash profiles/ab.sh r05b_dsplit_large "--model large-v2 --seconds 450 --steps 3 --warmup 1 --beam5-leg off" WHISPER_HIP_DECODER_SPLIT=0 WHISPER_HIP_DECODER_SPLIT=1
REPS=2 bash profiles/ab.sh r05b_dsplit_small "--model small --seconds 600 --steps 3 --warmup 1 --beam5-leg off" WHISPER_HIP_DECODER_SPLIT=0 WHISPER_HIP_DECODER_SPLIT=1
echo "[$(( $(date +%s) - T0 )) s] A/B done"
( cd /tmp && export TMPDIR=/tmp && timeout 400 python $R/bench.py --steps 40 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err )
python - $OUT/bench_default.json <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print("bench:", o["value"], "x,", o["ms_per_step"], "ms/step; from host PCM:", o["from_host_pcm"]["value"] if o.get("from_host_pcm") else None)
    print("roofline:", {k: o["roofline"][k] for k in ("kernel", "frac", "avg_launch_us")})
    b = o["beam5"]; print("beam5:", b["value"], b["ms_per_step"], b["config"]["generated_tokens_per_window"], b["config"]["stages_profiled_pass"])
    for k in (b["config"]["kernels"] or [])[:8]: print("   ", k["kernel"], k["launches_timed"], k["avg_launch_us"], k["share_of_decode_kernel_time"])
    l = o["large_v2"]; print("large_v2:", l["value"], l["ms_per_step"], l["stages"], l["step_ms"])
    for k in (l["kernels"] or [])[:8]: print("   ", k["kernel"], k["launches_timed"], k["avg_launch_us"], k["frac_of_hbm_peak"])
    print("mel:", o["mel_frontend"]["value"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] bench done"
# K12 counters on the large-v2 encoder (depth 4: the decode is short, the encoder is the run)
B="$R/bench.py --model large-v2 --seconds 120 --max-depth 4 --steps 1 --warmup 1 --large-v2-leg off --beam5-leg off --no-cpu-baseline"
( cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_k12a -o pmc -- python $B > $OUT/k12_pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d /tmp/p_k12b -o pmc -- python $B > $OUT/k12_pmc_b.log 2>&1
python $R/profiles/summarize_counters.py $(find /tmp/p_k12a /tmp/p_k12b -name '*.db') 2>&1 | grep -E "^==|gemm_f16x3|attention_f32|layernorm" | head -60 > $OUT/k12_counters.txt )
head -40 $OUT/k12_counters.txt
echo "[$(( $(date +%s) - T0 )) s] K12 counters done"
cd /tmp && export TMPDIR=/tmp
timeout 1150 python -m pytest $R/tests -m gpu -v -rA --durations=25 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -15
grep -E "^(small|large-v2|base.en)[ :]" $OUT/pytest_gpu.log | tail -14
echo "[$(( $(date +%s) - T0 )) s] suite done"
