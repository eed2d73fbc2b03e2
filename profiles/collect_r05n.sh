#!/bin/bash
# Round 5: sanity run of the rebuilt shipped library (header comments changed since the verification): every GPU test but the
# multi-minute batch-mode file, and smoke().
set -u
R=$PWD; OUT=$R/gpurun_out/r05n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 100 python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 420 python -m pytest $R/tests -x -q -m gpu -p no:cacheprovider --ignore=$R/tests/test_gpu_batchmode.py > $OUT/pytest_gpu_subset.log 2>&1
tail -3 $OUT/pytest_gpu_subset.log
