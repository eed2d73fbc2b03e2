#!/bin/bash
# Round 3, first GPU call: the whole -m gpu suite (incl. the new batch-mode / budget parity tests), the bench line, and
# the batch-mode profile of large-v2 on the current build.  Run through gpurun from the repo root:
#   bash profiles/collect_r03a.sh
set -u
R=$PWD
OUT=$R/gpurun_out/r03a
mkdir -p "$OUT"
python -m pytest tests -m gpu -q -s --durations=15 > "$OUT/pytest_gpu.log" 2>&1
tail -5 "$OUT/pytest_gpu.log"
python bench.py --steps 50 --warmup 5 > "$OUT/bench_tiny.log" 2>&1
grep '^{"metric' "$OUT/bench_tiny.log" > "$OUT/bench_tiny_en_30s.json"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" --model large-v2 --seconds 450 --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 > "$OUT/bench_large_v2.log" 2>&1
grep '^{"metric' "$OUT/bench_large_v2.log" > "$OUT/bench_large_v2_450s.json"
rm -rf /tmp/p_lv2
rocprofv3 --kernel-trace --stats -d /tmp/p_lv2 -o kt -- python "$R/bench.py" --model large-v2 --seconds 450 --steps 1 --warmup 1 \
  --no-cpu-baseline --mel-windows 8 > "$OUT/stats_large_v2.log" 2>&1
python "$R/profiles/summarize_rocprof.py" "$(find /tmp/p_lv2 -name '*.db' | head -1)" "$OUT/kernel_stats_large_v2_450s.csv"
ls -la "$OUT"
