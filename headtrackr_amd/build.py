"""Builds libheadtrackr_hip.so (HIP kernels + C ABI, gfx950 only) and the N-API addon in-tree with hipcc / g++.

    python -m headtrackr_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off: the pyramid and the stage sums must round every
binary64 operation exactly like the reference JS (no fused multiply-add).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libheadtrackr_hip.so")
ADDON = os.path.join(HERE, "js", "headtrackr_hip.node")

HIP_SOURCES = ["ht_context.hip", "ht_pyramid.hip", "ht_scan.hip", "ht_camshift.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
    "-I", os.path.join(ROOT, "include"), "-I", CSRC,
]
# The product build takes NO knobs from the environment: variants (instrumented / experimental kernels) are built by
# tools/build_alt.py with explicit -D arguments into alt/, never into the product library.


# per-source flags.  ht_camshift.hip: MachineLICM hoists the constant tables of the once-per-call epilogue (atan2 / sqrt polynomials, 40+
# VGPRs) out of k_cs_track_fused<true>'s call loop and the register allocator then SPILLS them (24 VGPRs, 100 B of scratch per lane) at
# the 128-VGPR cap of a 1024-thread workgroup; without the pass the kernel needs 122 VGPRs and no scratch (tests/test_abi.py checks the
# code object).
EXTRA_FLAGS = {"ht_camshift.hip": ["-mllvm", "-disable-machine-licm"]}


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))]
    deps += [os.path.join(ROOT, "include", "headtrackr_hip.h"), os.path.abspath(__file__)]
    if force or _newer(LIB, deps):
        objs, procs = [], []
        for s in srcs:  # the translation units are independent: compile them side by side (the whole library: ~6 s)
            o = os.path.splitext(s)[0] + ".o"
            if force or _newer(o, deps):
                cmd = [HIPCC, *HIP_FLAGS, *EXTRA_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o]
                if verbose:
                    print(" ".join(cmd))
                procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(o)
        for cmd, p in procs:
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, cmd)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_addon(force: bool = False, verbose: bool = False) -> str | None:
    src = os.path.join(CSRC, "ht_napi.cc")
    if not os.path.exists(src) or not os.path.exists("/usr/include/node/node_api.h"):
        return None
    deps = [src, os.path.join(ROOT, "include", "headtrackr_hip.h")]
    if force or _newer(ADDON, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", "/usr/include/node", "-I", os.path.join(ROOT, "include"),
               "-DNAPI_VERSION=7", "-DNODE_GYP_MODULE_NAME=headtrackr_hip", src, "-o", ADDON, "-L", HERE, "-lheadtrackr_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,--no-as-needed"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return ADDON


def build_all(force: bool = False, verbose: bool = False):
    lib = build_lib(force, verbose)
    addon = build_addon(force, verbose)
    return lib, addon


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
