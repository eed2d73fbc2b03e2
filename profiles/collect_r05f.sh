#!/bin/bash
# Round 5, GPU call 6: the profiles the bench line cites -- rocprofv3 kernel statistics + the two --pmc passes (FETCH_SIZE,
# WRITE_SIZE; separate runs, --kernel-trace only) of the bench step (tiny.en 30 s) and of the large-v2 450 s step.
set -u
R=$PWD; OUT=$R/gpurun_out/r05f; mkdir -p $OUT
T0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --large-v2-leg off --beam5-leg off --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o kt -- python $B --steps 5 --warmup 2 > "$OUT/stats.log" 2>&1
DB=$(find /tmp/p_stats -name '*.db' | head -1)
python "$R/profiles/summarize_rocprof.py" "$DB" "$OUT/kernel_stats_tiny_en_30s.csv"
python "$R/profiles/timeline_gaps.py" "$DB" > "$OUT/timeline_tiny_en_30s.txt" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o pmc -- python $B --steps 2 --warmup 1 > "$OUT/pmc_$C.log" 2>&1
done
python "$R/profiles/summarize_pmc.py" "$(find /tmp/p_FETCH_SIZE -name '*.db' | head -1)" \
  "$(find /tmp/p_WRITE_SIZE -name '*.db' | head -1)" "$OUT/pmc_traffic_tiny_en_30s.csv" "$OUT/pmc_traffic_tiny_en_30s.json"
head -6 "$OUT/kernel_stats_tiny_en_30s.csv"; head -c 600 "$OUT/pmc_traffic_tiny_en_30s.json"; echo
echo "[$(( $(date +%s) - T0 )) s] tiny profiles done"
BL="$R/bench.py --model large-v2 --seconds 450 --large-v2-leg off --beam5-leg off --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d /tmp/pl_stats -o kt -- python $BL --steps 2 --warmup 1 > "$OUT/stats_large.log" 2>&1
DBL=$(find /tmp/pl_stats -name '*.db' | head -1)
python "$R/profiles/summarize_rocprof.py" "$DBL" "$OUT/kernel_stats_large_v2_450s.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pl_$C -o pmc -- python $BL --steps 1 --warmup 1 > "$OUT/pmc_large_$C.log" 2>&1
done
python "$R/profiles/summarize_pmc.py" "$(find /tmp/pl_FETCH_SIZE -name '*.db' | head -1)" \
  "$(find /tmp/pl_WRITE_SIZE -name '*.db' | head -1)" "$OUT/pmc_traffic_large_v2_450s.csv" "$OUT/pmc_traffic_large_v2_450s.json"
head -12 "$OUT/kernel_stats_large_v2_450s.csv"; head -c 1500 "$OUT/pmc_traffic_large_v2_450s.json"; echo
echo "[$(( $(date +%s) - T0 )) s] large-v2 profiles done"
