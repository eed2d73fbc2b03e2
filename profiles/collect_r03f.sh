#!/bin/bash
# Round 3: the two-pass key ring (30 s window geometry, C = 1500) -- parity on the three decode paths, bench on each;
# A/B of the L2 warm-up experiment (libwhisper_hip_exp_l2warm.so) on the default bench.
#   bash profiles/collect_r03f.sh   (through gpurun)
set -u
R=$PWD
OUT=$R/gpurun_out/r03f
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_switches.py -q -k "whisper30 or 30_s_window" > "$OUT/pytest_whisper30.log" 2>&1; tail -3 "$OUT/pytest_whisper30.log"
B="--steps 20 --warmup 4 --no-cpu-baseline --mel-windows 8"
timeout 300 python bench.py --geometry whisper30 $B 2>&1 | grep '^{"metric' > "$OUT/bench_whisper30_persist.json"
WHISPER_HIP_PERSIST=0 timeout 300 python bench.py --geometry whisper30 $B 2>&1 | grep '^{"metric' > "$OUT/bench_whisper30_chain.json"
WHISPER_HIP_FUSE_X=0 timeout 300 python bench.py --geometry whisper30 $B 2>&1 | grep '^{"metric' > "$OUT/bench_whisper30_chunked.json"
for i in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_tiny_base_$i.json"
  WHISPER_HIP_LIB=$R/whisper-burn_amd/lib/libwhisper_hip_exp_l2warm.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --mel-windows 8 2>&1 | grep '^{"metric' > "$OUT/bench_tiny_l2warm_$i.json"
done
WHISPER_HIP_LIB=$R/whisper-burn_amd/lib/libwhisper_hip_exp_l2warm.so WHISPER_HIP_PS_STAMPS=/tmp/ps_l2warm.bin timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --mel-windows 8 > /dev/null 2>&1
python profiles/ps_timeline.py /tmp/ps_l2warm.bin > "$OUT/ps_timeline_l2warm.txt" 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03f/bench_*.json")):
    try:
        j = json.load(open(f)); print(f, j["value"], j["ms_per_step"], j["stages"]["decode_ms_per_step"], j["stages"].get("decode_kernels_per_token"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat "$OUT/ps_timeline_l2warm.txt"
