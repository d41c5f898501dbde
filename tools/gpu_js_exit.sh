#!/bin/bash
# does tests/js/bench_host.js exit cleanly — also while another process of the box holds HIP contexts (the pytest situation)?  Backtrace of the first crash.
python - <<PY
import numpy as np, sys, subprocess, os
sys.path.insert(0,'.')
from headtrackr_amd import synth
from headtrackr_amd.api import Context
synth.mixed_batch(64,320,240,seed0=1234).tofile('/tmp/c2.raw')
np.stack([synth.face_frame(320,240,[(90+2*k,50+k,96)]) for k in range(30)]).tofile('/tmp/track.raw')
c = Context(); c.detect_raw(synth.mixed_batch(4,320,240,seed0=1))   # this process keeps a HIP context, like pytest does
bad = 0
for i in range(8):
    r = subprocess.run(["node","tests/js/bench_host.js","0.3","/tmp/c2.raw","64","/tmp/track.raw","30"],capture_output=True,text=True)
    if r.returncode != 0:
        bad += 1
print("with a second HIP process alive:", bad, "of 8 runs exited non-zero")
if bad:
    r = subprocess.run(["/opt/rocm/bin/rocgdb","-batch","-ex","handle SIGUSR1 nostop noprint","-ex","run","-ex","bt 25","--args","node","tests/js/bench_host.js","0.3","/tmp/c2.raw","64","/tmp/track.raw","30"],capture_output=True,text=True)
    out = r.stdout
    i = out.find("received signal")
    print(out[i-200:i+3500] if i >= 0 else "no crash under gdb this time")
PY
