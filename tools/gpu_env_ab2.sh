#!/bin/bash
# A/B of one run-time knob: tools/gpu_env_ab2.sh VAR "v1 v2 ..." [workloads]  (3 repeats each, interleaved)
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"; }
VAR=$1; VALS=$2; WLS=${3:-"c2 c4"}
for rep in 1 2 3; do for wl in $WLS; do for v in $VALS; do
  ST=300; [ $wl = c4 ] && ST=80
  env $VAR=$v python bench.py --workload $wl --steps $ST --cpu-seconds 0 --no-sub 2>/dev/null | pr "$VAR=$v $wl"
done; done; done
