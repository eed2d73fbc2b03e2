#!/bin/bash
# Round 5, GPU call 1.  Variants are built in the build container first (tools/build_exp.sh; they travel with the snapshot):
#   base = the library of commit 8d133da (round-4 kernels + ADVICE fixes)   -> lib/libwhisper_hip_base.so
#   default = + every global access of the persistent kernel's role bodies through global_load instead of flat_load
#   nt_et / nt_w / nt_both = nt policy on the E^T tile stream / the layer weights + cached K,V / both
#   mel_pk / mel_fmac / mel_pkfmac, k12_term = the variants prepared at the end of round 4
# 1. persistent-kernel A/B on the default bench workload   2. mel variants (parity + frontend leg)
# 3. K12 issue order on large-v2                            4. the whole -m gpu suite (new fixtures, new order)
set -u
R=$PWD; OUT=$R/gpurun_out/r05a; mkdir -p $OUT
L=whisper-burn_amd/lib
T0=$(date +%s)
REPS=2 bash profiles/ab.sh r05a_ps "--steps 40 --warmup 3 --large-v2-leg off --beam5-leg off" $L/libwhisper_hip_base.so $L/libwhisper_hip.so $L/libwhisper_hip_exp_nt_et.so $L/libwhisper_hip_exp_nt_w.so $L/libwhisper_hip_exp_nt_both.so
echo "[$(( $(date +%s) - T0 )) s] persistent A/B done"
for rep in 1 2; do
  for v in libwhisper_hip.so libwhisper_hip_exp_mel_pk.so libwhisper_hip_exp_mel_fmac.so libwhisper_hip_exp_mel_pkfmac.so; do
    [ -f $R/$L/$v ] || continue
    WHISPER_HIP_LIB=$R/$L/$v timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --large-v2-leg off --beam5-leg off > $OUT/mel_${v}_$rep.json 2>/dev/null
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
  [ -f $R/$L/$v ] || continue
  echo "parity with $v:"; WHISPER_HIP_LIB=$R/$L/$v timeout 120 python -m pytest $R/tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "prep_audio or mel" 2>&1 | tail -2
done )
echo "[$(( $(date +%s) - T0 )) s] mel done"
REPS=2 bash profiles/ab.sh r05a_k12 "--model large-v2 --seconds 450 --steps 2 --warmup 1 --beam5-leg off" $L/libwhisper_hip.so $L/libwhisper_hip_exp_k12_term.so
grep -h -o '"encoder_ms_per_step": [0-9.]*' gpurun_out/ab_r05a_k12/variant*_rep*.log 2>/dev/null
echo "[$(( $(date +%s) - T0 )) s] K12 done"
cd /tmp && export TMPDIR=/tmp
timeout 1150 python -m pytest $R/tests -m gpu -v -rA --durations=25 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -15
grep -E "^(small|large-v2|base.en)[ :]" $OUT/pytest_gpu.log | tail -12
echo "[$(( $(date +%s) - T0 )) s] suite done"
