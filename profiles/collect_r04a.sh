#!/bin/bash
# Round 4, first GPU call: (1) stage-split numerics diagnostic at large-v2 / small under the decode switches,
# (2) A/B of the pairwise skinny GEMM on large-v2 450 s.   gpurun --timeout 1100 -- 'bash profiles/collect_r04a.sh'
set -u
R=$PWD
OUT=$R/gpurun_out/r04a
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
D=$R/whisper-burn_amd/tools/diag_stage_split.py
for model in large-v2 small; do
  c=/tmp/diag_$model.npz
  timeout 500 python $D $model $c base 36 > $OUT/diag_${model}_base.log 2>&1
  WHISPER_HIP_BATCH_SKINNY=0 timeout 200 python $D $model $c skinny0 36 > $OUT/diag_${model}_skinny0.log 2>&1
  WHISPER_HIP_CROSS_STREAM=0 timeout 200 python $D $model $c chunked 36 > $OUT/diag_${model}_chunked.log 2>&1
  WHISPER_HIP_SK_PAIR=1 timeout 200 python $D $model $c pair 36 > $OUT/diag_${model}_pair.log 2>&1
  WHISPER_HIP_ENCODER_SPLIT=1 timeout 300 python $D $model $c split 36 > $OUT/diag_${model}_split.log 2>&1
  grep -h "SUMMARY" -A 8 $OUT/diag_${model}_*.log | grep -v "^--"
done
cd $R
REPS=2 bash profiles/ab.sh r04a_skpair "--model large-v2 --seconds 450 --steps 1 --warmup 1" WHISPER_HIP_SK_PAIR=0 WHISPER_HIP_SK_PAIR=1
