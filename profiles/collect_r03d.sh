#!/bin/bash
# Round 3, final build: the whole -m gpu suite, the bench line, kernel statistics, the two PMC passes (HBM traffic of the
# persistent decode kernel) and the role timeline.   bash profiles/collect_r03d.sh   (through gpurun, from the repo root)
set -u
R=$PWD
OUT=$R/gpurun_out/r03d
mkdir -p "$OUT"
python -m pytest tests -m gpu -q -s --durations=12 > "$OUT/pytest_gpu.log" 2>&1
tail -4 "$OUT/pytest_gpu.log"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" --steps 40 --warmup 5 > "$OUT/bench.log" 2>&1
grep '^{"metric' "$OUT/bench.log" > "$OUT/bench_tiny_en_30s.json"
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o kt -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --mel-windows 8 \
  > "$OUT/stats.log" 2>&1
DB=$(find /tmp/p_stats -name '*.db' | head -1)
python "$R/profiles/summarize_rocprof.py" "$DB" "$OUT/kernel_stats_tiny_en_30s.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 \
    > "$OUT/pmc_$C.log" 2>&1
done
python "$R/profiles/summarize_pmc.py" "$(find /tmp/p_FETCH_SIZE -name '*.db' | head -1)" \
  "$(find /tmp/p_WRITE_SIZE -name '*.db' | head -1)" "$OUT/pmc_traffic_tiny_en_30s.csv" "$OUT/pmc_traffic_tiny_en_30s.json"
WHISPER_HIP_PS_STAMPS=$OUT/stamps.bin python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --mel-windows 8 > "$OUT/bench_stamps.log" 2>&1
python "$R/profiles/ps_timeline.py" "$OUT/stamps.bin" > "$OUT/ps_timeline.txt" 2>&1
rm -f "$OUT/stamps.bin"
ls -la "$OUT"; cat "$OUT/kernel_stats_tiny_en_30s.csv" | head -8; tail -c 1500 "$OUT/bench_tiny_en_30s.json"
