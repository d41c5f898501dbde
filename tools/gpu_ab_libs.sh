#!/bin/bash
# tools/gpu_ab_libs.sh NAME[=path.so] ... — A/B of whole library builds on ONE box: per-kernel device times with one batch in flight
# (tools/gpu_gen_times.py) and the bench's own frames/s for c2 and c4.  "cur" = the in-tree library; other names are alt/NAME.so.
OUT=gpurun_out; mkdir -p $OUT
LIB=headtrackr_amd/libheadtrackr_hip.so
cp $LIB /tmp/ab_cur.so
for spec in "$@"; do
  name=${spec%%=*}; path=alt/$name.so; [ "$name" = cur ] && path=/tmp/ab_cur.so; [[ "$spec" == *=* ]] && path=${spec#*=}
  [ -f "$path" ] || { echo "== $name: $path missing"; continue; }
  cp $path $LIB
  echo "== $name"
  [ -n "${CHECK:-}" ] && { timeout 300 python -m pytest tests/test_gpu_detect.py -q -x --no-header -p no:cacheprovider -k "pyramid" 2>&1 | tail -1; }
  for wl in ${WLS:-c2 c4}; do
    timeout 300 python tools/gpu_gen_times.py $wl 2> $OUT/ab_${name}_$wl.err | tr '\n' ' '; echo
    timeout 300 python bench.py --workload $wl --cpu-seconds 0 --no-sub --steps ${STEPS:-300} 2>> $OUT/ab_${name}_$wl.err | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   bench $wl', j['value'], 'frames/s', j['ms_per_step'], 'ms/step', j.get('ms_per_step_min'), j.get('ms_per_step_max'))"
  done
done
cp /tmp/ab_cur.so $LIB
