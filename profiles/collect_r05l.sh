#!/bin/bash
# Round 5, last GPU call: reference numbers of the other BASELINE configs on the final build, MFMA-busy counters of the two
# split-precision kernels, and the whole -m gpu suite once more on a fresh box (as the driver runs it).
set -u
R=$PWD; OUT=$R/gpurun_out/r05l; mkdir -p $OUT
T0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
run() { timeout 200 python $R/bench.py "$@" --no-cpu-baseline --large-v2-leg off --beam5-leg off 2>/dev/null | grep '^{"metric"' | tail -1; }
run --geometry whisper30 --steps 40 --warmup 3 > $OUT/bench_whisper30.json
run --model small --seconds 600 --steps 5 --warmup 2 > $OUT/bench_small_600s.json
run --model base.en --beam 5 --steps 10 --warmup 2 > $OUT/bench_base_en_beam5.json
python - $OUT/bench_whisper30.json $OUT/bench_small_600s.json $OUT/bench_base_en_beam5.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        o = json.loads(open(f).read())
        print(f.split("/")[-1], o["value"], "x", o["ms_per_step"], "ms/step;", o["config"]["workload"][:90], "| tokens", o["config"]["tokens_out"])
    except Exception as e:
        print(f, "failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] configs done"
B="$R/bench.py --model large-v2 --seconds 120 --max-depth 8 --steps 1 --warmup 1 --large-v2-leg off --beam5-leg off --no-cpu-baseline"
WHISPER_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o pmc -- python $B > $OUT/pmc_mfma.log 2>&1
python $R/profiles/summarize_counters.py $(find /tmp/p_mfma -name '*.db') 2>&1 | grep -E "^==|skinny|gemm_f16x3|cross_attn_stream" | head -20 > $OUT/mfma_counters_large_v2.txt
cat $OUT/mfma_counters_large_v2.txt
echo "[$(( $(date +%s) - T0 )) s] counters done"
timeout 1100 python -m pytest $R/tests -x -q -m gpu -rA --durations=10 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -6
echo "[$(( $(date +%s) - T0 )) s] suite done"
