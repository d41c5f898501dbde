#!/bin/bash
# tools/gpu_sweep_env.sh VAR "v1 v2 ..." [workloads] — A/B of one environment knob on the bench's own numbers (frames/s, ms per step)
VAR=$1; VALS=$2; WLS=${3:-"c2 c4"}
for wl in $WLS; do
  for v in $VALS; do
    if [ "$v" = "default" ]; then unset $VAR; else export $VAR=$v; fi
    python bench.py --workload $wl --cpu-seconds 0 --no-sub --steps ${STEPS:-400} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$wl $VAR=$v', j['value'], j['ms_per_step'], j['ms_per_step_min'], j['ms_per_step_max'], j['kernel_ms_per_step'])"
  done
done
