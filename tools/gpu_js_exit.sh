#!/bin/bash
# does tests/js/bench_host.js exit cleanly? (Node 12 + N-API finalizers at environment teardown crashed it; 10 runs)
python - <<PY
import numpy as np, sys
sys.path.insert(0,'.')
from headtrackr_amd import synth
synth.mixed_batch(64,320,240,seed0=1234).tofile('/tmp/c2.raw')
np.stack([synth.face_frame(320,240,[(90+2*k,50+k,96)]) for k in range(30)]).tofile('/tmp/track.raw')
PY
bad=0
for i in 1 2 3 4 5 6 7 8 9 10; do
  node tests/js/bench_host.js 0.2 /tmp/c2.raw 64 /tmp/track.raw 30 > /tmp/js_out.txt 2>/tmp/js_err.txt; rc=$?
  [ $rc -ne 0 ] && { bad=$((bad+1)); echo "run $i exit $rc"; }
done
echo "bench_host.js: $bad of 10 runs exited non-zero"
