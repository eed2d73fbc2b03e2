#!/bin/bash
# Round 6: the split-precision attention with 64-query blocks on small grids (tiny.en x 3 windows) against the exact-f32 key-split kernel.
# (WHISPER_HIP_ATTN_F16_SMALL existed only in the variant measured here; the change was reverted -- LABLOG R6.11.)
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --large-v2-leg off --beam5-leg off --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'encoder_ms', d['stages']['encoder_ms_per_step'], 'tokens_checked', d['config']['tokens_checked'])"; }
for i in 1 2; do
run WHISPER_HIP_ATTN_F16_SMALL=1
run WHISPER_HIP_ATTN_F16_SMALL=0
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py tests/test_gpu_e2e.py tests/test_gpu_budget.py -x -q 2>&1 | tail -25
