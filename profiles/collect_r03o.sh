#!/bin/bash
# Round 3: after the plane-store fix of the skinny GEMM (three / four row tiles): the ten tests that failed in r03n, the
# `small` bench again, and the split-precision encoder (WHISPER_HIP_ENCODER_SPLIT=1): parity at real shapes, then A/B.
set -u
R=$PWD
OUT=$R/gpurun_out/r03o
mkdir -p "$OUT"
timeout 900 python -m pytest -q -s \
  "tests/test_gpu_batchmode.py::test_small_10min_all_51_windows_depth_100_batch_mode" \
  "tests/test_gpu_edge.py" "tests/test_gpu_scale.py" "tests/test_gpu_session.py" \
  "tests/test_gpu_workloads.py::test_workload_tokens_match_oracle" \
  -k "small_10min or finishing or long_audio or many_windows" > "$OUT/pytest_refailed.log" 2>&1; tail -3 "$OUT/pytest_refailed.log"
WHISPER_HIP_ENCODER_SPLIT=1 timeout 600 python -m pytest -q -s tests/test_gpu_parity.py tests/test_gpu_workloads.py \
  -k "forward_encoder_micro or tiny_en_forward_real_shape or greedy_chain_and_logprobs_live or small_forward_real_shape or tiny_bench" \
  > "$OUT/pytest_split.log" 2>&1; tail -3 "$OUT/pytest_split.log"
B="--steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8"
timeout 600 python bench.py --model small --seconds 600 $B 2>&1 | grep '^{"metric' > "$OUT/bench_small_600s.json"
for S in 0 1; do
  WHISPER_HIP_ENCODER_SPLIT=$S timeout 600 python bench.py --model large-v2 --seconds 450 --steps 3 --warmup 1 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_large_v2_450s_split$S.json"
done
WHISPER_HIP_ENCODER_SPLIT=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_tiny_split1.json"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03o/bench_*.json")):
    try:
        j = json.load(open(f)); s = j["stages"]
        print(f, j["value"], j["ms_per_step"], "enc", round(s["encoder_ms_per_step"], 2), "ckv", round(s["cross_kv_ms_per_step"], 2), "dec", round(s["decode_ms_per_step"], 1), "enc TF", s.get("encoder_TFLOPs_algorithmic"))
    except Exception as e:
        print(f, "unreadable", e)
PY
