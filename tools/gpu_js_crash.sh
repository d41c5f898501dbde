#!/bin/bash
# one-off: does tests/js/bench_host.js exit cleanly? (exit code + backtrace if not)
python - <<PY
import numpy as np, sys
sys.path.insert(0,'.')
from headtrackr_amd import synth
synth.mixed_batch(64,320,240,seed0=1234).tofile('/tmp/c2.raw')
np.stack([synth.face_frame(320,240,[(90+2*k,50+k,96)]) for k in range(30)]).tofile('/tmp/track.raw')
PY
node tests/js/bench_host.js 0.3 /tmp/c2.raw 64 /tmp/track.raw 30 > gpurun_out/js_bench.out 2> gpurun_out/js_bench.err; echo "node exit $?"
cut -c1-300 gpurun_out/js_bench.out; tail -5 gpurun_out/js_bench.err
