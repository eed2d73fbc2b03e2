#!/bin/bash
# Round 5, GPU call 4: beam search (9 - 16 live rows) on the fused sublayer path -- A/B against batch mode on the bench's
# depth-100 beam-5 leg, then every beam / session / switch test.
set -u
R=$PWD; OUT=$R/gpurun_out/r05d; mkdir -p $OUT
T0=$(date +%s)
REPS=2 bash profiles/ab.sh r05d_fuse16 "--steps 10 --warmup 2 --large-v2-leg off" WHISPER_HIP_FUSE16=0 WHISPER_HIP_FUSE16=1
REPS=2 bash profiles/ab.sh r05d_fuse16_base "--model base.en --steps 10 --warmup 2 --large-v2-leg off" WHISPER_HIP_FUSE16=0 WHISPER_HIP_FUSE16=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_r05d_fuse16*/variant*_rep*.log")):
    try:
        o = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
        b = o["beam5"]
        print(f.split("/")[-2][8:], f.split("/")[-1], "greedy", o["value"], "beam5", b["value"], b["ms_per_step"], "ms;", b["config"]["generated_tokens_per_window"], b["config"]["stages_profiled_pass"]["launches"], "launches")
        for k in b["config"]["kernels"][:6]: print("      ", k["kernel"][:50], k["launches_timed"], k["avg_launch_us"])
    except Exception as e:
        print(f, "failed", e)
PY
echo "[$(( $(date +%s) - T0 )) s] A/B done"
cd /tmp && export TMPDIR=/tmp
timeout 700 python -m pytest $R/tests/test_gpu_golden.py $R/tests/test_gpu_workloads.py $R/tests/test_gpu_session.py $R/tests/test_gpu_switches.py $R/tests/test_gpu_scale.py $R/tests/test_gpu_edge.py $R/tests/test_legacy_modes.py -m gpu -q -p no:cacheprovider \
  -k "not large and not small_10min" 2>&1 | tail -12 | tee $OUT/pytest_beam.log
echo "[$(( $(date +%s) - T0 )) s] tests done"
