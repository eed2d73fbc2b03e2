for rep in 1 2; do for v in base ld16 ld12; do
  if [ $v = base ]; then L=$PWD/whisper-burn_amd/lib/libwhisper_hip.so; else L=$PWD/whisper-burn_amd/lib/libwhisper_hip_exp_$v.so; fi
  WHISPER_HIP_LIB=$L timeout 300 python bench.py --large-v2-leg off --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['beam5']; lg=[k for k in b['config']['kernels'] if 'logits' in k['kernel']][0]
print('$v rep $rep', 'beam5', b['value'], b['ms_per_step'], 'logits us', lg['avg_launch_us'], b['config']['tokens_checked'])"
done; done
