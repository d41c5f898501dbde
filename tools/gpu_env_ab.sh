#!/bin/bash
# tools/gpu_env_ab.sh VAR v1 v2 ... — bench c2/c4 with an environment knob at several values (default build)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
VAR=$1; shift
summ() { python -c "
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], 'fps',d['value'],'ms/step',d['ms_per_step'],'kernels',d.get('kernel_ms_per_step'))
" $1 "$2"; }
if [ "${AB_TESTS:-1}" = 1 ]; then
  echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
fi
for v in "$@"; do
  for wl in ${AB_WL:-c2 c4}; do
    ST=10; [ $wl = c4 ] && ST=4
    env $VAR=$v timeout 300 python bench.py --workload $wl --steps $ST --warmup 2 --cpu-seconds 0 > $OUT/env_${v}_$wl.json 2>$OUT/env_${v}_$wl.err; summ $OUT/env_${v}_$wl.json "$VAR=$v $wl"
  done
done
