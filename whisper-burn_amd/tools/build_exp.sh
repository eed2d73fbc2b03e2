#!/bin/bash
# An experimental variant of the library next to the product one: the given sources recompiled with extra -D flags, the
# other objects reused.  For A/B runs on the GPU box (WHISPER_HIP_LIB=<variant> python bench.py ...); never the default.
#   whisper-burn_amd/tools/build_exp.sh l2warm "-DWB_EXP_L2WARM" decode_persist.hip
set -eu
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../csrc"
make -j8 > /dev/null
mkdir -p build_exp_$NAME
OBJS=""
for o in build/*.o; do
  b=$(basename "$o" .o)
  use=$o
  for s in "$@"; do
    if [ "$b" = "$s" ]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -c "$s" -o "build_exp_$NAME/$b.o"
      use="build_exp_$NAME/$b.o"
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/libwhisper_hip_exp_$NAME.so $OBJS -lz
echo "../lib/libwhisper_hip_exp_$NAME.so"
