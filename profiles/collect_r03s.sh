#!/bin/bash
# Round 3, last build: smoke(), the split-precision parity subset, the default bench line.
set -u
R=$PWD
OUT=$R/gpurun_out/r03s
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
WHISPER_HIP_ENCODER_SPLIT=1 timeout 200 python -m pytest -q -s tests/test_gpu_parity.py tests/test_gpu_workloads.py \
  -k "forward_encoder_micro or tiny_en_forward_real_shape or greedy_chain_and_logprobs_live or small_forward_real_shape or tiny_bench" \
  > "$OUT/pytest_split.log" 2>&1; tail -2 "$OUT/pytest_split.log"
timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_tiny_en_30s.json"
python -c "import json; j=json.load(open('$OUT/bench_tiny_en_30s.json')); print(j['value'], j['ms_per_step'])"
