#!/bin/bash
# Round 3: where the first phase of the attention roles goes (developer build with finer stamps).
set -u
R=$PWD
OUT=$R/gpurun_out/r03g
mkdir -p "$OUT"
WHISPER_HIP_LIB=$R/whisper-burn_amd/lib/libwhisper_hip_exp_fine.so WHISPER_HIP_PS_STAMPS=/tmp/ps_fine.bin timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --mel-windows 8 > /dev/null 2>&1
python profiles/ps_fine.py /tmp/ps_fine.bin > "$OUT/ps_fine.txt" 2>&1
cat "$OUT/ps_fine.txt"
