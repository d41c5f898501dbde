#!/bin/bash
# tools/gpu_ab_bands.sh — k_resample_bands (option rs_bands=1) against k_resample on one box: parity of every plane, then device ms per step
# and wall ms per step, two repetitions each, C2 three batches in flight / 720p two.
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_detect.py -m gpu -q --no-header -p no:cacheprovider -x -k "both_generation_kernels or pyramid" > $OUT/pytest_bands.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/pytest_bands.log
for rep in 1 2; do
  for o in "" "rs_bands=1" ${EXTRA_OPTS:-}; do
    timeout 120 python tools/gpu_kernel_times.py c2 "$o" 3 2>/dev/null | tail -1
    timeout 120 python tools/gpu_kernel_times.py c4 "$o" 2 2>/dev/null | tail -1
  done
done | tee $OUT/ab_bands.txt
