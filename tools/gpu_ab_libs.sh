#!/bin/bash
# tools/gpu_ab_libs.sh "LIB|OPTIONS" ...   — A/B of library builds and context options on one box: every entry is timed with
# tools/gpu_kernel_times.py at C2 (three batches in flight) and 720p (two), REPS repetitions interleaved.  LIB = "-" (the product
# library) or a name under alt/ (tools/build_alt.py).  PARITY=1 first runs the pyramid parity tests with every entry's library.
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
REPS=${REPS:-2}
for rep in $(seq $REPS); do
  for e in "$@"; do
    lib=${e%%|*}; opt=${e#*|}
    if [ "$lib" = "-" ]; then unset HEADTRACKR_HIP_LIB; else export HEADTRACKR_HIP_LIB=$PWD/alt/$lib.so; fi
    if [ "${PARITY:-0}" = 1 ] && [ $rep = 1 ]; then
      HT_TEST_OPTIONS="$opt" timeout 600 python -m pytest tests/test_gpu_detect.py -m gpu -q --no-header -p no:cacheprovider -x -k "pyramid" 2>&1 | tail -1
    fi
    for wl in ${WLS:-c2 c4}; do
      d=3; [ $wl = c4 ] && d=2
      echo -n "[$lib|$opt] "; timeout 120 python tools/gpu_kernel_times.py $wl "$opt" $d 2>/dev/null | tail -1
    done
  done
done | tee $OUT/ab_libs.txt
