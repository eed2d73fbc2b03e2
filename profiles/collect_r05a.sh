#!/bin/bash
# First GPU call of round 5 (prepared at the end of round 4, when the GPU budget was spent).  Build the variants FIRST, in the
# build container (they travel with the snapshot):
#   cd whisper-burn_amd/csrc
#   bash ../tools/build_exp.sh mel_pk     "-DWB_MEL_PK_DFT" mel.hip
#   bash ../tools/build_exp.sh mel_fmac   "-DWB_MEL_FMAC_ASM" mel.hip
#   bash ../tools/build_exp.sh mel_pkfmac "-DWB_MEL_PK_DFT -DWB_MEL_FMAC_ASM" mel.hip
#   bash ../tools/build_exp.sh k12_term   "-DWB_F16X3_TERM_MAJOR" gemm_f16x3.hip
# then:  gpurun --timeout 1500 -- 'bash profiles/collect_r05a.sh'
# 1. mel variants: parity (test_gpu_parity -k "prep_audio or mel") + the frontend leg of the bench, two alternating repetitions
# 2. K12 issue order: large-v2 450 s, base vs term-major, two repetitions (+ the encoder's own milliseconds)
# 3. the whole -m gpu suite on the DEFAULT build (it last ran before the mel tap-loop change of round 4)
set -u
R=$PWD; OUT=$R/gpurun_out/r05a; mkdir -p $OUT
L=$R/whisper-burn_amd/lib
T0=$(date +%s)
for rep in 1 2; do
  for v in libwhisper_hip.so libwhisper_hip_exp_mel_pk.so libwhisper_hip_exp_mel_fmac.so libwhisper_hip_exp_mel_pkfmac.so; do
    [ -f $L/$v ] || continue
    WHISPER_HIP_LIB=$L/$v timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --large-v2-leg off --beam5-leg off > $OUT/mel_${v}_$rep.json 2>/dev/null
    python - "$OUT/mel_${v}_$rep.json" "$v" <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print(sys.argv[2], "mel G frames/s", round(o["mel_frontend"]["value"] / 1e9, 3))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  done
done
( cd /tmp && export TMPDIR=/tmp
for v in libwhisper_hip_exp_mel_pk.so libwhisper_hip_exp_mel_pkfmac.so; do
  [ -f $L/$v ] || continue
  echo "parity with $v:"; WHISPER_HIP_LIB=$L/$v timeout 120 python -m pytest $R/tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "prep_audio or mel" 2>&1 | tail -2
done )
echo "[$(( $(date +%s) - T0 )) s] mel done"
[ -f $L/libwhisper_hip_exp_k12_term.so ] && REPS=2 bash profiles/ab.sh r05a_k12 "--model large-v2 --seconds 450 --steps 2 --warmup 1" whisper-burn_amd/lib/libwhisper_hip.so whisper-burn_amd/lib/libwhisper_hip_exp_k12_term.so
grep -h -o '"encoder_ms_per_step": [0-9.]*' gpurun_out/ab_r05a_k12/variant*_rep*.log 2>/dev/null
echo "[$(( $(date +%s) - T0 )) s] K12 done"
cd /tmp && export TMPDIR=/tmp
timeout 1100 python -m pytest $R/tests -m gpu -v -rA --durations=10 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -12
echo "[$(( $(date +%s) - T0 )) s] suite done"
