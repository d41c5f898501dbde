#!/usr/bin/env python3
"""tools/build_alt.py NAME DEFINE[=V] ...   build an instrumented variant of the library into alt/NAME.so (objects in a temp dir, the
                                           product library is not touched) and record the hash of the sources it was built from;
tools/build_alt.py --check NAME            exit 1 unless alt/NAME.so was built from the sources as they are NOW.
tools/gpu_final.sh refuses a stale variant: round 2 committed "timelines" that were the tracebacks of stale builds."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from headtrackr_amd import build as B  # noqa: E402

ALT = os.path.join(ROOT, "alt")


def source_hash():
    h = hashlib.sha256()
    files = [os.path.join(B.CSRC, f) for f in sorted(os.listdir(B.CSRC)) if f.endswith((".hip", ".h", ".inc"))] + [os.path.join(ROOT, "include", "headtrackr_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def check(name):
    info = os.path.join(ALT, name + ".json")
    so = os.path.join(ALT, name + ".so")
    if not (os.path.exists(info) and os.path.exists(so)):
        return False
    return json.load(open(info)).get("source_sha256") == source_hash()


if __name__ == "__main__":
    if sys.argv[1] == "--check":
        ok = check(sys.argv[2])
        print(f"alt/{sys.argv[2]}.so: {'current' if ok else 'STALE or missing'}")
        raise SystemExit(0 if ok else 1)
    name, defs = sys.argv[1], sys.argv[2:]
    os.makedirs(ALT, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        objs, procs = [], []
        for s in B.HIP_SOURCES:
            o = os.path.join(td, os.path.splitext(s)[0] + ".o")
            cmd = [B.HIPCC, *B.HIP_FLAGS, *B.EXTRA_FLAGS.get(s, []), *[f"-D{d}" if "=" in d else f"-D{d}=1" for d in defs], "-c", os.path.join(B.CSRC, s), "-o", o]
            procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
            objs.append(o)
        for cmd, p in procs:
            _, err = p.communicate()
            if p.returncode != 0:
                sys.stderr.write(err)
                raise SystemExit(f"build failed: {' '.join(cmd)}")
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ALT, name + ".so"), *objs, "-ldl"])
    json.dump({"defines": defs, "source_sha256": source_hash()}, open(os.path.join(ALT, name + ".json"), "w"))
    print(f"alt/{name}.so built with {defs}")
