"""Fingerprint of the gfx950 code objects inside libheadtrackr_hip.so — what ties the committed PMC constants (profiles/traffic.json) to a build.

The library is a fat binary: one clang offload bundle per translation unit, each holding that unit's gfx950 code object.  `code_objects()`
finds the bundles in the file's bytes (no external tool: the bundle header is magic + entry table), hashes the machine code (.text) of every gfx950 entry and names it
after the kernels it contains.  tools/gpu_pmc.sh records the fingerprint of the library it profiles, tools/collect_profiles.py stores it as
traffic.json's `_build`, and `stale_units()` tells bench.py whether the library it is timing still is that build."""
import hashlib
import os
import struct

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# translation unit -> a kernel only it contains
UNITS = {"pyramid": b"k_gray_linear", "scan": b"k_scan_tiles", "camshift": b"k_cs_hist"}
# which units a workload's counters depend on
WORKLOAD_UNITS = {"c2": ("pyramid", "scan"), "c4": ("pyramid", "scan"), "c3": ("pyramid", "scan", "camshift"), "c5": ("pyramid", "scan", "camshift")}


def default_lib():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.environ.get("HEADTRACKR_HIP_LIB") or os.path.join(root, "headtrackr_amd", "libheadtrackr_hip.so")


def code_objects(lib=None):
    """{unit: first 16 hex digits of sha256(gfx950 code object)} for the units in UNITS; {} if the file cannot be read"""
    try:
        data = open(lib or default_lib(), "rb").read()
    except OSError:
        return {}
    out = {}
    pos = data.find(MAGIC)
    while pos >= 0:
        try:
            (n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
            p = pos + len(MAGIC) + 8
            for _ in range(min(n, 16)):
                off, size, tlen = struct.unpack_from("<QQQ", data, p)
                triple = data[p + 24:p + 24 + tlen]
                p += 24 + tlen
                if b"gfx950" in triple and size:
                    co = data[pos + off:pos + off + size]
                    for unit, marker in UNITS.items():
                        if marker in co:
                            out[unit] = hashlib.sha256(_machine_code(co)).hexdigest()[:16]
        except struct.error:
            pass
        pos = data.find(MAGIC, pos + 1)
    return out


def _machine_code(co):
    """The .text and .rodata sections of a code object (ELF64): the instructions, the kernel descriptors and the constant tables.  The whole object would not do: hipcc derives the names of
    anonymous-namespace kernels from a hash of the SOURCE FILE'S PATH (-cuid), so the same sources built in another directory give other
    symbol and metadata bytes — and the same machine code.  Falls back to the whole object if the section table cannot be read."""
    try:
        if co[:4] != b"\x7fELF" or co[4] != 2:
            return co
        shoff, = struct.unpack_from("<Q", co, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", co, 0x3A)
        def sh(i):
            name, _typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", co, shoff + i * shentsize)
            return name, off, size
        _n, stroff, strsize = sh(shstrndx)
        names = co[stroff:stroff + strsize]
        code = b""
        for i in range(shnum):
            name, off, size = sh(i)
            if names[name:names.index(b"\0", name)] in (b".text", b".rodata"):  # instructions + constant tables (kernel descriptors live in .rodata)
                code += co[off:off + size]
        if code:
            return code
    except (struct.error, ValueError):
        pass
    return co


def stale_units(recorded, workload, lib=None):
    """units of `workload` whose code object differs from the recorded fingerprint (or is missing on either side); None = nothing recorded"""
    if not recorded:
        return None
    now = code_objects(lib)
    return [u for u in WORKLOAD_UNITS.get(workload, ()) if not now.get(u) or now.get(u) != recorded.get(u)]
