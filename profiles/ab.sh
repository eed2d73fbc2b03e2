#!/bin/bash
# A/B of two builds or two switch settings on ONE MI355X box (boxes of the pool differ by ~8 %, and a box can have a slow
# first minute -- see r02_f_ab_batch_mode.txt -- so variants alternate and every variant runs REPS times):
#   gpurun --timeout 200 -- 'bash profiles/ab.sh <tag> "<bench.py flags>" <variant> [<variant> ...]'
# A variant is either a library path (lib/libwhisper_hip_base.so: a copy of the previous build kept next to the new
# one -- *.so files travel with the snapshot) or ENV=VALUE settings separated by commas (WHISPER_HIP_CROSS_STREAM=0).
# A run of the batch-mode configs takes ~8 s (small, 600 s of audio, 3 steps), the default bench ~6 s without the CPU leg.
set -u
TAG=$1; FLAGS=$2; shift 2
R=$PWD
OUT=$R/gpurun_out/ab_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPS=${REPS:-2}
T0=$(date +%s)
for rep in $(seq 1 "$REPS"); do
  i=0
  for v in "$@"; do
    i=$((i + 1))
    log="$OUT/variant${i}_rep${rep}.log"
    if [ -f "$R/$v" ]; then
      WHISPER_HIP_LIB="$R/$v" timeout 120 python "$R/bench.py" $FLAGS --no-cpu-baseline > "$log" 2>&1
    else
      env $(echo "$v" | tr ',' ' ') timeout 120 python "$R/bench.py" $FLAGS --no-cpu-baseline > "$log" 2>&1
    fi
    echo "[$(( $(date +%s) - T0 )) s] variant $i ($v) rep $rep rc=$?: $(grep -o '"value": [0-9.]*' "$log" | head -1) $(grep -o '"ms_per_step": [0-9.]*' "$log" | head -1)" \
      | tee -a "$OUT/summary.txt"
  done
done
