run() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | grep "aggressor\|mismatching\|passed\|failed" | cut -c1-300; }
run PROBE_MODE=persistent PROBE_SAME_LEN=1 python whisper-burn_amd/tools/probe_threads_enc.py 300
run PROBE_MODE=fresh python whisper-burn_amd/tools/probe_threads_enc.py 300
run WHISPER_HIP_GPU_TURN=0 PROBE_MODE=persistent PROBE_SAME_LEN=1 python whisper-burn_amd/tools/probe_threads_enc.py 300
run python whisper-burn_amd/tools/probe_threads.py 1 150
run python whisper-burn_amd/tools/probe_threads.py 3 150
run WHISPER_HIP_GPU_TURN=0 python whisper-burn_amd/tools/probe_threads.py 1 150
for i in 1 2 3 4 5; do run python -m pytest tests/test_gpu_concurrency.py -q -x; done
