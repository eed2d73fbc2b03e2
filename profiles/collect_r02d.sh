#!/bin/bash
# Round 2, last collection (12 GPU-minutes were left): ONE gpurun call, most important evidence first, every step
# under its own timeout, everything written under gpurun_out/r02d/ as it is produced.
#   gpurun --timeout 660 -- 'bash profiles/collect_r02d.sh'
set -u
R=$PWD
OUT=$R/gpurun_out/r02d
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T0=$(date +%s)
BUDGET=${BUDGET:-570}
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a "$OUT/timeline.log"; }
left() { echo $(( BUDGET - ($(date +%s) - T0) )); }
run() {   # run <seconds> <log> <cmd...>: skipped when less than that much time is left
  local t=$1 log=$2; shift 2
  if [ "$(left)" -lt "$t" ]; then stamp "SKIP (only $(left) s left, needs $t): $*"; return 99; fi
  timeout "$t" "$@" > "$OUT/$log" 2>&1
  local rc=$?
  stamp "rc=$rc $log"
  return $rc
}
stamp start
# 1. the tests this session added or touched (opt-in 30 s geometry, prompt conditioning) + the workload rows incl. the new one
run 200 pytest_new.log python -m pytest "$R/tests/test_gpu_edge.py::test_whisper_window_geometry_is_opt_in" \
    "$R/tests/test_legacy_modes.py" -m gpu -q -p no:cacheprovider
run 200 pytest_workloads.log python -m pytest "$R/tests/test_gpu_workloads.py::test_workload_tokens_match_oracle" -m gpu -q \
    -p no:cacheprovider -k "tiny_bench or tiny_whisper30 or base_beam5 or tiny_beam5"
# 2. the bench line (default = the judged configuration) with the CPU leg
run 150 bench_default.log python "$R/bench.py" --steps 100 --warmup 5
grep '^{"metric' "$OUT/bench_default.log" > "$OUT/bench_tiny_en_30s.json" 2>/dev/null
# 3. the opt-in perf geometry (one 29.9 s window + its 3 s tail)
run 120 bench_whisper30.log python "$R/bench.py" --geometry whisper30 --steps 60 --warmup 3
grep '^{"metric' "$OUT/bench_whisper30.log" > "$OUT/bench_tiny_en_whisper30.json" 2>/dev/null
# 4. kernel statistics of both (rocprofv3 --kernel-trace --stats only)
if [ "$(left)" -gt 90 ]; then
  timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o kt -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline \
    > "$OUT/stats_default.log" 2>&1
  stamp "rc=$? rocprof default"
  DB=$(find /tmp/p_stats -name '*.db' | head -1)
  [ -n "$DB" ] && python "$R/profiles/summarize_rocprof.py" "$DB" "$OUT/kernel_stats_tiny_en_30s.csv" >> "$OUT/stats_default.log" 2>&1
fi
if [ "$(left)" -gt 90 ]; then
  timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/p_stats30 -o kt -- python "$R/bench.py" --geometry whisper30 --steps 5 --warmup 2 \
    --no-cpu-baseline > "$OUT/stats_whisper30.log" 2>&1
  stamp "rc=$? rocprof whisper30"
  DB=$(find /tmp/p_stats30 -name '*.db' | head -1)
  [ -n "$DB" ] && python "$R/profiles/summarize_rocprof.py" "$DB" "$OUT/kernel_stats_tiny_en_whisper30.csv" >> "$OUT/stats_whisper30.log" 2>&1
fi
# 5. whatever time is left: the whole GPU suite (the driver re-runs it at round end anyway)
L=$(left)
if [ "$L" -gt 60 ]; then
  timeout $(( L - 15 )) python -m pytest "$R/tests" -m gpu -q -x -p no:cacheprovider --durations=15 > "$OUT/pytest_all.log" 2>&1
  stamp "rc=$? pytest all"
fi
stamp done
tail -n 5 "$OUT"/pytest_new.log "$OUT"/pytest_workloads.log
tail -c 700 "$OUT/bench_tiny_en_30s.json"; echo
tail -c 500 "$OUT/bench_tiny_en_whisper30.json"; echo
cat "$OUT/timeline.log"
