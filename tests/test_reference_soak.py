"""A short run of tools/cpu_soak_reference.py: random geometries / frame families / tracker set-ups through the REFERENCE JS itself and
through the oracle (and, on the reference's output, through the product's host-side code: ht_group_rects, the JS facade's grouping,
Smoother, headposition; random facetrackr / Tracker / main.js sequences through the unchanged facade on the oracle-backed mock addon).  Only where the reference is present (this container; the GPU box has no /root/reference) — the long run's
summary is profiles/r05_reference_soak.txt."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference/headtrackr.js"


@pytest.mark.skipif(not os.path.exists(REF) or shutil.which("node") is None, reason="needs /root/reference and node")
@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_and_host_code_equal_the_reference_on_random_cases(seed):
    from test_js_host import _build_oracle_addon

    if os.path.exists("/usr/include/node/node_api.h"):
        _build_oracle_addon()  # with it the soak also drives the JS facade's state machines on the oracle-backed mock addon
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_soak_reference.py"), "4", str(seed)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("oracle vs reference JS soak") and "all identical" in last and "MISMATCH" not in r.stdout, last
    if os.path.exists(os.path.join(ROOT, "tests", "js", "oracle_addon.node")):
        assert last.endswith("all passed") and "mock addon" in last, last
    assert " 0 detect cases" not in last
