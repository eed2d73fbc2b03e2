#!/bin/bash
# Round 4, fourth GPU call: the hand-off stress output, the tests the r04c call cut off, A/B of the prompt prefill folded into
# the persistent launch, kernel statistics + PMC traffic of the bench step.   gpurun --timeout 900 -- 'bash profiles/collect_r04d.sh'
set -u
R=$PWD
OUT=$R/gpurun_out/r04d
mkdir -p "$OUT"
T0=$(date +%s)
timeout 120 whisper-burn_amd/lib/handoff_stress 1100 > $OUT/handoff_stress_1100.txt 2>&1; echo "rc=$?" >> $OUT/handoff_stress_1100.txt
timeout 200 whisper-burn_amd/lib/handoff_stress 20000 > $OUT/handoff_stress_20000.txt 2>&1; echo "rc=$?" >> $OUT/handoff_stress_20000.txt
cat $OUT/handoff_stress_1100.txt $OUT/handoff_stress_20000.txt
echo "[$(( $(date +%s) - T0 )) s] stress done"
REPS=3 bash profiles/ab.sh r04d_prefill "--steps 100 --warmup 5 --large-v2-leg off --beam5-leg off" WHISPER_HIP_PERSIST_PREFILL=0 WHISPER_HIP_PERSIST_PREFILL=1
echo "[$(( $(date +%s) - T0 )) s] A/B done"
cd /tmp && export TMPDIR=/tmp
timeout 420 python -m pytest $R/tests/test_gpu_shard_rccl.py $R/tests/test_gpu_switches.py $R/tests/test_gpu_workloads.py -m gpu -q -rA -p no:cacheprovider -k "not small_10min and not large_window" 2>&1 | tail -45 > $OUT/pytest_a.log
tail -30 $OUT/pytest_a.log
echo "[$(( $(date +%s) - T0 )) s] tests done"
# kernel statistics + HBM traffic of the bench step (the legs that are not the headline are off under the profiler)
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
head -8 "$OUT/kernel_stats_tiny_en_30s.csv"; cat "$OUT/pmc_traffic_tiny_en_30s.json" | head -c 1500
echo "[$(( $(date +%s) - T0 )) s] profiles done"
