run() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | grep "aggressor\|mismatching" | cut -c1-300; }
for v in "" SETTLE NOSLP NO_PK; do
  L=whisper-burn_amd/lib/libwhisper_hip_exp_mel_$v.so; [ -z "$v" ] && L=whisper-burn_amd/lib/libwhisper_hip.so
  run PROBE_MODE=persistent PROBE_SAME_LEN=1 WHISPER_HIP_LIB=$L python whisper-burn_amd/tools/probe_threads_enc.py 300
done
