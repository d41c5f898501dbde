#!/usr/bin/env python3
"""tools/kernel_resources.py [lib.so] — register / LDS / scratch figures of every gfx950 kernel in the built library, read from the
code objects' metadata (llvm-readelf --notes), no GPU needed.  tests/test_abi.py asserts on it (no kernel spills); the table is
committed as profiles/rNN_kernel_resources.txt."""
import os
import re
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
        "max_flat_workgroup_size")


def kernel_resources(lib=None):
    lib = lib or os.path.join(ROOT, "headtrackr_amd", "libheadtrackr_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{BIN}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        for i, st in enumerate(starts):  # one bundle per translation unit
            part = os.path.join(td, f"b{i}.bin")
            open(part, "wb").write(data[st: starts[i + 1] if i + 1 < len(starts) else len(data)])
            co = os.path.join(td, f"co{i}.o")
            subprocess.check_call([f"{BIN}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"],
                                  stderr=subprocess.DEVNULL)
            notes = subprocess.run([f"{BIN}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = None  # a kernel's entry starts with `- .agpr_count:` (keys are sorted); its own .name comes in the middle
            for ln in notes.splitlines():
                m = re.match(r"\s*(-?)\s*\.(\w+):\s*(.*)$", ln)
                if not m:
                    continue
                dash, k, v = m.group(1), m.group(2), m.group(3).strip().strip("'")
                if dash and k == "agpr_count":
                    cur = {}
                if cur is None:
                    continue
                if k == "name" and (v.startswith("_Z") or v.startswith("k_")):
                    out[v] = cur
                elif k in KEYS:
                    cur[k] = int(v)
    return out


def short(name):
    d = name
    for tool in (f"{BIN}/llvm-cxxfilt", "c++filt"):
        try:
            d = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip() or name
            break
        except Exception:
            continue
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    return re.sub(r"^void ", "", d.split("(")[0])


if __name__ == "__main__":
    res = kernel_resources(sys.argv[1] if len(sys.argv) > 1 else None)
    print(f"{'kernel':44s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s} {'wg':>5s}")
    for n, r in sorted(res.items(), key=lambda kv: short(kv[0])):
        if "vgpr_count" not in r:
            continue
        print(f"{short(n):44s} {r.get('vgpr_count', 0):5d} {r.get('sgpr_count', 0):5d} {r.get('vgpr_spill_count', 0):6d} {r.get('sgpr_spill_count', 0):6d} "
              f"{r.get('private_segment_fixed_size', 0):7d} {r.get('group_segment_fixed_size', 0):6d} {r.get('max_flat_workgroup_size', 0):5d}")
