#!/bin/bash
# tools/gpu_ab_host.sh NAME... — A/B of library builds whose HOST side differs: host time of one batch's post-processing, the 20-step block
# the driver times, and the long run.
LIB=headtrackr_amd/libheadtrackr_hip.so
cp $LIB /tmp/ab_cur.so
for name in "$@"; do
  path=alt/$name.so; [ "$name" = cur ] && path=/tmp/ab_cur.so
  cp $path $LIB; echo "== $name"
  timeout 100 python tools/gpu_host_post.py 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/gpu_fill_drain.py 20 2 2>&1 | grep -v amdgpu.ids
  for k in 20 20 1000; do
    python bench.py --no-sub --cpu-seconds 0 --steps $k --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('  bench K=$k',d['value'],d['ms_per_step'],d['ms_per_step_min'],d['ms_per_step_max'])"
  done
done
cp /tmp/ab_cur.so $LIB
