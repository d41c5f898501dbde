#!/bin/bash
# one-off: backtrace of the exit-time crash of tests/js/bench_host.js (rocgdb, up to 6 runs)
python - <<PY
import numpy as np, sys
sys.path.insert(0,'.')
from headtrackr_amd import synth
synth.mixed_batch(64,320,240,seed0=1234).tofile('/tmp/c2.raw')
np.stack([synth.face_frame(320,240,[(90+2*k,50+k,96)]) for k in range(30)]).tofile('/tmp/track.raw')
PY
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex "thread apply all bt 12" --args node tests/js/bench_host.js 0.3 /tmp/c2.raw 64 /tmp/track.raw 30 > gpurun_out/js_gdb_$i.txt 2>&1
  if grep -q "SIGSEGV\|SIGABRT" gpurun_out/js_gdb_$i.txt; then echo "crash in run $i"; grep -n "SIGSEGV\|SIGABRT" gpurun_out/js_gdb_$i.txt | head -3; grep -A14 "^Thread 1 \|received signal" gpurun_out/js_gdb_$i.txt | head -60; break; else echo "run $i clean"; fi
done
