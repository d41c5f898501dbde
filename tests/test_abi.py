"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares,
and the ctypes mirrors have the C struct sizes.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from headtrackr_amd import native


@pytest.fixture(scope="module")
def built_lib():
    from headtrackr_amd import build

    build.build_lib()
    return native.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "headtrackr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ht_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(built_lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(built_lib, s), f"libheadtrackr_hip.so does not export {s}"
    assert sorted(native.SYMBOLS) == syms, "native.SYMBOLS is out of sync with include/headtrackr_hip.h"


def test_abi_version_and_struct_sizes(built_lib):
    assert built_lib.ht_abi_version() == 1
    assert C.sizeof(native.Config) == 32
    assert native.HIT_DTYPE.itemsize == 24
    assert native.RECT_DTYPE.itemsize == 48


def test_create_without_gpu_fails_loudly(built_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from headtrackr_amd.api import Context, HtError

    with pytest.raises(HtError) as e:
        Context()
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)


def test_generated_cascade_code_is_in_sync(tmp_path):
    """headtrackr_amd/csrc/ht_cascade_gen.inc is generated from data/cascade.bin by tools/gen_cascade_code.py (default
    arguments: 8 stages, 28 loads per group); a stale copy would silently fall back to the table-driven kernels because
    of its FNV guard — or worse, encode other thresholds.  Regenerate and compare."""
    import subprocess
    import sys

    out = tmp_path / "gen.inc"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_cascade_code.py"), "8", "28", str(out)], stdout=subprocess.DEVNULL)
    want = open(os.path.join(ROOT, "headtrackr_amd", "csrc", "ht_cascade_gen.inc")).read()
    assert out.read_text() == want
