#!/usr/bin/env python3
"""tools/disasm.py KERNEL_SUBSTRING [lib.so] — gfx950 disassembly of one kernel of the built library (llvm-objdump on the code object
extracted from the fat binary), with an instruction-class count; no GPU needed."""
import collections
import os
import re
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def disasm(name, lib=None):
    lib = lib or os.path.join(ROOT, "headtrackr_amd", "libheadtrackr_hip.so")
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{BIN}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        for i, st in enumerate(starts):
            part, co = os.path.join(td, f"b{i}.bin"), os.path.join(td, f"co{i}.o")
            open(part, "wb").write(data[st: starts[i + 1] if i + 1 < len(starts) else len(data)])
            subprocess.check_call([f"{BIN}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"],
                                  stderr=subprocess.DEVNULL)
            txt = subprocess.run([f"{BIN}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            blocks = re.split(r"\n(?=[0-9a-f]+ <)", txt)
            for b in blocks:
                head = b.split("\n", 1)[0]
                if name in head and ">:" in head:
                    return b
    return None


if __name__ == "__main__":
    b = disasm(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    if b is None:
        raise SystemExit("kernel not found")
    print(b)
    cnt = collections.Counter()
    for ln in b.splitlines()[1:]:
        m = re.match(r"\s+(\w+)", ln)
        if m:
            op = m.group(1)
            cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
            cnt[cls] += 1
    print("# static instruction counts:", dict(cnt), file=sys.stderr)
