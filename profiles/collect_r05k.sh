#!/bin/bash
# Round 5, GPU call 11: 9 - 16 live rows, the final fold + LayerNorm once (dec_fold_ln_rows) + a plain 16-row logits pass against
# the LN prologue in every logits block; bench's depth-100 beam-5 leg, then the beam tests.
set -u
R=$PWD; OUT=$R/gpurun_out/r05k; mkdir -p $OUT
T0=$(date +%s)
REPS=2 bash profiles/ab.sh r05k_fold16 "--steps 10 --warmup 2 --large-v2-leg off" WHISPER_HIP_LOGITS_FOLD16=0 WHISPER_HIP_LOGITS_FOLD16=1
REPS=1 bash profiles/ab.sh r05k_fold16_base "--model base.en --steps 10 --warmup 2 --large-v2-leg off" WHISPER_HIP_LOGITS_FOLD16=0 WHISPER_HIP_LOGITS_FOLD16=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_r05k_fold16*/variant*_rep*.log")):
    try:
        o = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
        b = o["beam5"]
        lg = [(k["kernel"][:22], k["avg_launch_us"]) for k in b["config"]["kernels"] if "logits" in k["kernel"] or "resolve" in k["kernel"]]
        print(f.split("/")[-2][8:], f.split("/")[-1], "beam5", b["value"], b["ms_per_step"], "ms;", lg)
    except Exception as e:
        print(f, "failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] A/B done"
cd /tmp && export TMPDIR=/tmp
timeout 400 python -m pytest $R/tests/test_gpu_golden.py $R/tests/test_gpu_workloads.py $R/tests/test_gpu_session.py $R/tests/test_gpu_scale.py -m gpu -q -p no:cacheprovider \
  -k "beam or session or golden" 2>&1 | tail -5 | tee $OUT/pytest_beam.log
echo "[$(( $(date +%s) - T0 )) s] tests done"
