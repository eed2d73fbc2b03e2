#!/bin/bash
# Round 4, mel variants: chunk-major filterbank (eight rows advance together) and launch bounds of three blocks per CU.
set -u
R=$PWD; OUT=$R/gpurun_out/r04h; mkdir -p $OUT
for rep in 1 2; do
for v in libwhisper_hip.so libwhisper_hip_exp_mel_fb.so libwhisper_hip_exp_mel_lb3.so libwhisper_hip_exp_mel_fblb3.so; do
  WHISPER_HIP_LIB=$R/whisper-burn_amd/lib/$v timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --large-v2-leg off --beam5-leg off > $OUT/${v}_$rep.json 2>/dev/null
  python - "$OUT/${v}_$rep.json" "$v" <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print(sys.argv[2], "mel frames/s", o["mel_frontend"]["value"], "ms/pass", o["mel_frontend"]["ms_per_pass"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
done
cd /tmp && export TMPDIR=/tmp
for v in libwhisper_hip_exp_mel_fb.so libwhisper_hip_exp_mel_fblb3.so; do
  WHISPER_HIP_LIB=$R/whisper-burn_amd/lib/$v timeout 120 python -m pytest $R/tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "prep_audio or mel" 2>&1 | tail -2
done
