"""The JavaScript host (the reference's own language): CPU-side checks of the facade here, GPU parity of the whole
JS -> N-API -> C ABI -> HIP chain against the reference-JS golden vectors on the GPU box."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT, load_golden
from headtrackr_amd import synth

NODE = shutil.which("node")


def _build():
    from headtrackr_amd import build

    build.build_all()


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_facade_exports_and_host_grouping():
    """no GPU needed: exports, cascade unpacking, addon symbols, grouping of golden raw hits"""
    _build()
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "facade_cpu.js")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out["errors"]
    assert out["abi"] == 1


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_js_host_parity_on_gpu(tmp_path):
    _build()
    det, cam, ft = load_golden("detect.json"), load_golden("camshift.json"), load_golden("facetrackr.json")
    cache = {}

    def ffile(gen, w, h):
        key = json.dumps([gen, w, h], sort_keys=True)
        if key not in cache:
            fn = f"f{len(cache)}.raw"
            synth.make(gen, w, h).tofile(str(tmp_path / fn))
            cache[key] = fn
        return cache[key]

    post = load_golden("post.json")
    dbg = load_golden("debug.json")
    job = {"detect": [], "camshift": [], "facetrackr": [], "pipeline": [], "mainjs": []}
    for c in det["cases"]:
        if c["w"] > 640:
            continue
        job["detect"].append(dict(name=c["name"], w=c["w"], h=c["h"], interval=3 if "interval3" in c["name"] else 5,
                                  frame=ffile(c["gen"], c["w"], c["h"]), golden={k: c[k] for k in ("whitebalance", "gray_rgba_crc", "raw", "grouped", "min_neighbors")}))
    for c in cam["cases"]:
        job["camshift"].append(dict(name=c["name"], w=c["w"], h=c["h"], frames=[ffile(g, c["w"], c["h"]) for g in c["gen"]],
                                    golden={k: c[k] for k in ("rect", "calcAngles", "calls", "backprojection_crc", "pdf_samples")}))
    for c in ft["cases"]:
        job["facetrackr"].append(dict(name=c["name"], w=c["w"], h=c["h"], frames=[ffile(g, c["w"], c["h"]) for g in c["gen"]],
                                      golden={k: c[k] for k in ("params", "calls", "events")}))
    for c in post["cases"]:
        if c["kind"] == "pipeline":
            g = {k: c[k] for k in ("params", "calls", "fov")}
            g["whitebalancing"] = False
            job["pipeline"].append(dict(name=c["name"], w=c["w"], h=c["h"], frames=[ffile(x, c["w"], c["h"]) for x in c["gen"]], golden=g))
    for c in dbg["cases"]:
        job["mainjs"].append(dict(name=c["name"], w=c["w"], h=c["h"], frames=[ffile(x, c["w"], c["h"]) for x in c["gen"]],
                                  golden={k: c[k] for k in ("params", "calls", "fov")}))
    jf = tmp_path / "job.json"
    jf.write_text(json.dumps(job))
    # (with one visible GPU the sharded call has one rank and skips RCCL; tests/test_gpu_shapes.py::test_allgather_best_faces_over_rccl
    #  forces the RCCL path — ncclCommInitAll alone takes ~50 s on the test box, once is enough)
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "parity_gpu.js"), str(jf)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out["errors"]
    assert out["checked"] > 500
