#!/bin/bash
# Round 6, the committed library: rocprofv3 kernel statistics of the default bench step (tiny.en 30 s, greedy + the beam-5 leg) --
# the same command as collect_r06.sh's first block.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --large-v2-leg off --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o kt -- python $B --steps 5 --warmup 2 > "$OUT/r06_x_stats.log" 2>&1
DB=$(find /tmp/p_stats -name '*.db' | head -1)
python "$R/profiles/summarize_rocprof.py" "$DB" "$OUT/r06_x_kernel_stats_tiny_en_30s.csv"
python "$R/profiles/timeline_gaps.py" "$DB" > "$OUT/r06_x_timeline_tiny_en_30s.txt" 2>&1
head -8 "$OUT/r06_x_kernel_stats_tiny_en_30s.csv"; tail -1 "$OUT/r06_x_stats.log" | cut -c1-600
