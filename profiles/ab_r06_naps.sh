for rep in 1 2; do for v in base napm1 napm2 napp1; do
  if [ $v = base ]; then L=$PWD/whisper-burn_amd/lib/libwhisper_hip.so; else L=$PWD/whisper-burn_amd/lib/libwhisper_hip_exp_$v.so; fi
  WHISPER_HIP_LIB=$L timeout 300 python bench.py --large-v2-leg off --beam5-leg off --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v rep $rep', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['tokens_checked'])"
done; done
