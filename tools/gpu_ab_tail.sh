#!/bin/bash
# A/B of the tail kernels and the pipeline depth at C2 / 720p on one box
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_detect.py -q -k "tail_kernels" -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for o in "rs_tailtable=0" "rs_tailtable=2" "rs_tailtable=1"; do
  timeout 120 python tools/gpu_kernel_times.py c2 $o 3 2>/dev/null | tail -1
done
done
for o in "rs_tailtable=1" "rs_tailtable=2" "rs_tailtable=0"; do
  timeout 120 python tools/gpu_kernel_times.py c4 $o 2 2>/dev/null | tail -1
done
for d in 3 4; do timeout 120 python tools/gpu_kernel_times.py c2 "" $d 2>/dev/null | tail -1; done
