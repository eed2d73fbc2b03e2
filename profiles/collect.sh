#!/bin/bash
# Collects the round's evidence on an MI355X box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag>        e.g.  gpurun -- 'bash profiles/collect.sh r01'
# Writes into gpurun_out/collect_<tag>/ (merged back by gpurun); the files to judge are then copied
# into profiles/ by hand.  Counter passes use --kernel-trace only (no other trace domain).
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/collect_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (with the CPU baseline leg)
python "$R/bench.py" --steps 40 --warmup 5 > "$OUT/bench.log" 2>&1
grep '^{"metric' "$OUT/bench.log" > "$OUT/bench_tiny_en_30s.json"
# 2. kernel statistics of the same command (no CPU leg: the profiler would only slow it down)
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o kt -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline \
  > "$OUT/stats.log" 2>&1
DB=$(find /tmp/p_stats -name '*.db' | head -1)
python "$R/profiles/summarize_rocprof.py" "$DB" "$OUT/kernel_stats_tiny_en_30s.csv"
python "$R/profiles/timeline_gaps.py" "$DB" > "$OUT/timeline_tiny_en_30s.txt" 2>&1
cp "$DB" "$OUT/kernel_trace.db" 2>/dev/null || true
# 3. HBM traffic counters: one pass per counter
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline \
    > "$OUT/pmc_$C.log" 2>&1
done
python "$R/profiles/summarize_pmc.py" "$(find /tmp/p_FETCH_SIZE -name '*.db' | head -1)" \
  "$(find /tmp/p_WRITE_SIZE -name '*.db' | head -1)" "$OUT/pmc_traffic_tiny_en_30s.csv" "$OUT/pmc_traffic_tiny_en_30s.json"
ls -la "$OUT"
tail -c 600 "$OUT/bench_tiny_en_30s.json"
