#!/bin/bash
# tools/gpu_c5_ab.sh [options...] — camshift / C5 parity tests, then the C5 bench line (8 feeds and 1 feed) for every ht_config.options string given
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c5.py tests/test_gpu_camshift.py tests/test_js_host.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest2.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest2.log
for o in "$@"; do for f in 8 1; do
  timeout 300 python bench.py --workload c5 --feeds $f --cpu-seconds 0 --options $o --no-sub 2>$OUT/c5_${o}_$f.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o feeds $f', d['value'], d['ms_per_step'], d.get('parity_exact'))"
done; done
