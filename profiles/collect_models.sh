#!/bin/bash
# Kernel statistics of BASELINE.json's batch-mode configs on the current build (run through gpurun from the repo root):
#   bash profiles/collect_models.sh <tag>
# config #4 small, 10 min (51 windows) and config #5's per-GPU share large-v2, 450 s (38 windows); + base.en beam 5.
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/collect_models_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench args...
  local name=$1; shift
  python "$R/bench.py" "$@" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_$name.log" 2>&1
  grep '^{"metric' "$OUT/bench_$name.log" > "$OUT/bench_$name.json"
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o kt -- python "$R/bench.py" "$@" --steps 1 --warmup 1 --no-cpu-baseline \
    > "$OUT/stats_$name.log" 2>&1
  python "$R/profiles/summarize_rocprof.py" "$(find /tmp/p_$name -name '*.db' | head -1)" "$OUT/kernel_stats_$name.csv"
}
run small_600s --model small --seconds 600
run large_v2_450s --model large-v2 --seconds 450
run base_en_30s_beam5 --model base.en --seconds 30 --beam 5
ls -la "$OUT"
