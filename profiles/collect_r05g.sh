#!/bin/bash
# Round 5, GPU call 7: the two --pmc passes of the large-v2 450 s step that crashed in call 6 (rocprofv3 counter collection +
# replayed hipGraphs of ~6 000 nodes): eager launches (WHISPER_HIP_GRAPH=0) and a short decode (max depth 10 -- the per-launch
# traffic of the weight-stream kernels does not depend on the depth).
set -u
R=$PWD; OUT=$R/gpurun_out/r05g; mkdir -p $OUT
T0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
export WHISPER_HIP_GRAPH=0
BL="$R/bench.py --model large-v2 --seconds 450 --max-depth 10 --large-v2-leg off --beam5-leg off --no-cpu-baseline"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pl_$C -o pmc -- python $BL --steps 1 --warmup 1 > "$OUT/pmc_large_$C.log" 2>&1
  echo "$C rc=$? [$(( $(date +%s) - T0 )) s]"; find /tmp/pl_$C -name '*.db' | head -2
done
python "$R/profiles/summarize_pmc.py" "$(find /tmp/pl_FETCH_SIZE -name '*.db' | head -1)" \
  "$(find /tmp/pl_WRITE_SIZE -name '*.db' | head -1)" "$OUT/pmc_traffic_large_v2_450s.csv" "$OUT/pmc_traffic_large_v2_450s.json"
head -c 2500 "$OUT/pmc_traffic_large_v2_450s.json"; echo
tail -3 "$OUT/pmc_large_FETCH_SIZE.log"
echo "[$(( $(date +%s) - T0 )) s] done"
