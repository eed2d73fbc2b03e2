#!/bin/bash
# Round 6: the profiles the bench line and DESIGN.md cite -- rocprofv3 kernel statistics of the default bench step (tiny.en 30 s,
# greedy + the beam-5 leg) and of the large-v2 450 s step, the two --pmc traffic passes of each (FETCH_SIZE, WRITE_SIZE: separate
# runs, --kernel-trace only; the large-v2 ones eager and at depth 10 -- counter collection + replayed graphs of ~6 000 nodes
# crashed in round 5, and the per-launch traffic of the weight-stream kernels does not depend on the depth), and the matrix-pipe
# counters of the large-v2 encoder (K12 with activation pieces, K5 on the 16-bit path).
set -u
R=$PWD; OUT=$R/gpurun_out/r06p; mkdir -p $OUT
T0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --large-v2-leg off --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o kt -- python $B --steps 5 --warmup 2 > "$OUT/stats.log" 2>&1
DB=$(find /tmp/p_stats -name '*.db' | head -1)
python "$R/profiles/summarize_rocprof.py" "$DB" "$OUT/kernel_stats_tiny_en_30s.csv"
python "$R/profiles/timeline_gaps.py" "$DB" > "$OUT/timeline_tiny_en_30s.txt" 2>&1
BG="$R/bench.py --large-v2-leg off --beam5-leg off --no-cpu-baseline"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o pmc -- python $BG --steps 2 --warmup 1 > "$OUT/pmc_$C.log" 2>&1
done
python "$R/profiles/summarize_pmc.py" "$(find /tmp/p_FETCH_SIZE -name '*.db' | head -1)" \
  "$(find /tmp/p_WRITE_SIZE -name '*.db' | head -1)" "$OUT/pmc_traffic_tiny_en_30s.csv" "$OUT/pmc_traffic_tiny_en_30s.json"
head -8 "$OUT/kernel_stats_tiny_en_30s.csv"; head -c 600 "$OUT/pmc_traffic_tiny_en_30s.json"; echo
echo "[$(( $(date +%s) - T0 )) s] tiny profiles done"
BL="$R/bench.py --model large-v2 --seconds 450 --large-v2-leg off --beam5-leg off --no-cpu-baseline"
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/pl_stats -o kt -- python $BL --steps 2 --warmup 1 > "$OUT/stats_large.log" 2>&1
DBL=$(find /tmp/pl_stats -name '*.db' | head -1)
python "$R/profiles/summarize_rocprof.py" "$DBL" "$OUT/kernel_stats_large_v2_450s.csv"
head -14 "$OUT/kernel_stats_large_v2_450s.csv"
echo "[$(( $(date +%s) - T0 )) s] large-v2 kernel stats done"
export WHISPER_HIP_GRAPH=0
BLS="$R/bench.py --model large-v2 --seconds 450 --max-depth 10 --large-v2-leg off --beam5-leg off --no-cpu-baseline"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pl_$C -o pmc -- python $BLS --steps 1 --warmup 1 > "$OUT/pmc_large_$C.log" 2>&1
  echo "$C rc=$? [$(( $(date +%s) - T0 )) s]"
done
python "$R/profiles/summarize_pmc.py" "$(find /tmp/pl_FETCH_SIZE -name '*.db' | head -1)" \
  "$(find /tmp/pl_WRITE_SIZE -name '*.db' | head -1)" "$OUT/pmc_traffic_large_v2_450s.csv" "$OUT/pmc_traffic_large_v2_450s.json"
head -c 1500 "$OUT/pmc_traffic_large_v2_450s.json"; echo
BK="$R/bench.py --model large-v2 --seconds 120 --max-depth 4 --steps 1 --warmup 1 --large-v2-leg off --beam5-leg off --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_k12a -o pmc -- python $BK > $OUT/k12_pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d /tmp/p_k12b -o pmc -- python $BK > $OUT/k12_pmc_b.log 2>&1
python $R/profiles/summarize_counters.py $(find /tmp/p_k12a /tmp/p_k12b -name '*.db') 2>&1 | grep -E "^==|gemm_f16x3|attention_f|layernorm" | head -60 > $OUT/k12_counters.txt
head -30 $OUT/k12_counters.txt
# the mel frontend's issue mix (what 1.29 G frames/s is a fraction of): dynamic VALU / LDS instruction counts and busy cycles
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/p_mela -o pmc -- python $BG --steps 1 --warmup 0 > $OUT/mel_pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES -d /tmp/p_melb -o pmc -- python $BG --steps 1 --warmup 0 > $OUT/mel_pmc_b.log 2>&1
python $R/profiles/summarize_counters.py $(find /tmp/p_mela /tmp/p_melb -name '*.db') 2>&1 | grep -E "^==|mel_spectrogram" | head -40 > $OUT/mel_counters.txt
cat $OUT/mel_counters.txt
echo "[$(( $(date +%s) - T0 )) s] done"
