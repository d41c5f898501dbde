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
    assert out["abi"] == 2


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_addon_argument_handling_never_crashes_or_swallows():
    """the product addon (csrc/ht_napi.cc) under malformed calls, no GPU needed: ~11 000 calls with too few / wrong arguments end in a
    JavaScript exception (never a crash, never a silent `undefined`: a run of this script found 21 entry points returning silently when
    called with too few arguments; a context handle and a device-buffer handle are told apart by a tag, not reinterpreted)"""
    _build()
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "addon_args.js")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])  # a crash would be a signal exit
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out["errors"]
    assert out["calls"] > 10000 and out["threw"] > 0.9 * out["calls"] - 700


def _parity_job(tmp_path):
    """the golden vectors of the reference JS as a job for tests/js/parity_gpu.js (frames written as raw RGBA files)"""
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
    return job


def _build_oracle_addon():
    """tests/js/oracle_addon.node: the CPU oracle as a Node addon (test infrastructure, compiled with the oracle's own flags)"""
    src, out = os.path.join(ROOT, "tests", "js", "oracle_addon.c"), os.path.join(ROOT, "tests", "js", "oracle_addon.node")
    dep = os.path.join(ROOT, "oracle", "ht_oracle.c")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(dep)):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-std=c11", "-D_GNU_SOURCE",
                               "-I/usr/include/node", src, "-o", out, "-lm"])
    return out


@pytest.mark.skipif(NODE is None or not os.path.exists("/usr/include/node/node_api.h"), reason="node / node_api.h not installed")
def test_js_facade_host_logic_on_the_cpu_mock(tmp_path):
    """The UNCHANGED JavaScript facade (headtrackr.js + tracker.js) against the reference-JS golden vectors without a GPU: the product
    addon is replaced by tests/js/mock_addon.js (its entry points on the CPU oracle), everything above the addon interface is the
    product's code — seq construction, grouping, getWhitebalance, camshift.Tracker with its debug getters, the facetrackr WB -> VJ -> CS
    state machine with its events, the headtrackr.Tracker loop (status events, Smoother, headposition), the main.js debug overlay
    (stroke calls + canvas pixels), and the host side of detect_objects_batch (async, sharded + gathered) and ccv.DeviceBatch (detect,
    detectBest over three contexts with re-enqueue, fused whitebalance, trackSequence, pinned ingest) — the WHOLE job of the GPU run,
    at every golden frame size."""
    _build_oracle_addon()
    job = _parity_job(tmp_path)
    job["cpu_mock"] = True
    jf = tmp_path / "job.json"
    jf.write_text(json.dumps(job))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "parity_cpu.js"), str(jf)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out["errors"]
    assert out["checked"] > 2000
    exact, total = (int(v) for v in out["cs_parity"].split("/"))
    assert total > 0 and exact == total
    calls = out["addon_calls"]  # the facade really went through the addon interface
    for k in ("createContext", "setGeometry", "upload", "grayscale", "detect", "detectEnqueue", "detectCollect", "whitebalance",
              "whitebalanceBound", "camshiftReserve", "camshiftInitBound", "camshiftTrackBound", "detectAsync", "allgatherBest",
              "deviceAlloc", "deviceUpload", "bindDevice", "collectBest", "detectWhitebalance", "camshiftTrackSequence", "uploadAsync",
              "swapFrames", "hostAlloc", "hostFree", "deviceFree", "destroy"):
        assert calls.get(k, 0) > 0, (k, calls)
    # every frame size was announced through setGeometry (V8-computed level sizes, headtrackr.js levelDims): the addon never had to
    # re-build a geometry on its own (this run found getWhitebalance skipping that step)
    assert calls.get("implicitGeometry", 0) == 0, calls


@pytest.mark.skipif(NODE is None or not os.path.exists("/usr/include/node/node_api.h"), reason="node / node_api.h not installed")
def test_c5_loop_from_node_on_the_cpu_mock(cascade, tmp_path):
    """tests/js/c5_stream.js — the C5 loop of the JavaScript host (ccv.DeviceBatch: resident frame sets, detect step enqueued behind the
    outstanding track steps, initTracker on the floored best faces, enqueue-only track steps collected from the ring) — UNCHANGED, on the
    oracle-backed mock addon: 2 feeds of 1920x1080, 35 steps (detect at steps 0 and 30).  Every step's result must be what the reference's
    per-feed loop produces (oracle): best faces, floored rects, every track object in order.  This checks the host-side sequencing — which
    set is bound when, which step a collected result belongs to; the kernels are checked by tests/test_gpu_c5.py on the GPU."""
    import math

    import numpy as np

    from oracle import ht_oracle as ho

    _build_oracle_addon()
    W, H, nuniq, K, steps = 1920, 1080, 8, 2, 35
    uniq = synth.stream_feed_frames(nuniq, W, H, 0)
    raw, outf = tmp_path / "uniq.raw", tmp_path / "out.json"
    uniq.tofile(str(raw))
    r = subprocess.run([NODE, "-r", os.path.join(ROOT, "tests", "js", "mock_preload.js"), os.path.join(ROOT, "tests", "js", "c5_stream.js"), "parity",
                        str(raw), str(nuniq), str(K), str(steps), str(outf)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    os.unlink(str(raw))
    got = json.loads(outf.read_text())
    assert len(got["steps"]) == steps and got["feeds"] == K
    oracles, tracked = [None] * K, 0
    for i, s in enumerate(got["steps"]):
        assert s["step"] == i
        if i % 30 == 0:
            best = np.array(s["best"]).reshape(K, 6)
            for f in range(K):
                u = synth.stream_frame_index(i, f, nuniq)
                wb = ho.best_faces(uniq[u : u + 1], cascade.blob, 1)[0]
                assert list(best[f]) == [wb["x"], wb["y"], wb["width"], wb["height"], wb["confidence"], float(wb["neighbors"])], (i, f)
                rect = tuple(int(math.floor(v)) for v in best[f][:4])
                assert tuple(s["rects"][4 * f : 4 * f + 4]) == rect
                oracles[f] = ho.Camshift(True)
                oracles[f].init_tracker(uniq[u], rect)
        else:
            t = np.array(s["track"]).reshape(K, 9)
            for f in range(K):
                sw, to = oracles[f].track(uniq[synth.stream_frame_index(i, f, nuniq)])
                assert [int(v) for v in t[f][5:9]] == list(sw), (i, f)
                assert [t[f][0], t[f][1], t[f][2], t[f][3]] == [to["x"], to["y"], to["width"], to["height"]], (i, f)
                assert abs(t[f][4] - to["angle"]) <= 1e-12, (i, f)
                tracked += 1
    assert tracked == K * (steps - 2)


@pytest.mark.skipif(NODE is None or not os.path.exists("/usr/include/node/node_api.h"), reason="node / node_api.h not installed")
def test_js_host_benchmark_script_on_the_cpu_mock(tmp_path):
    """tests/js/bench_host.js (bench.py's js_host sub-record) unchanged on the oracle-backed mock addon: both batch paths see the same
    faces, and the drop-in facetrackr.Tracker ends where the unmodified reference JS ends on the same frames (where oracle/_ref exists).
    Speeds mean nothing here; the GPU run of the same script is tests/test_js_host.py::test_js_host_benchmark_runs."""
    import numpy as np

    _build_oracle_addon()
    W, H, n, nt = 320, 240, 16, 30
    c2, tr = tmp_path / "c2.raw", tmp_path / "track.raw"
    synth.mixed_batch(n, W, H, seed0=1234).tofile(str(c2))
    np.stack([synth.face_frame(W, H, [(90 + 2 * k, 50 + k, 96)]) for k in range(nt)]).tofile(str(tr))
    r = subprocess.run([NODE, "-r", os.path.join(ROOT, "tests", "js", "mock_preload.js"), os.path.join(ROOT, "tests", "js", "bench_host.js"), "0.1",
                        str(c2), str(n), str(tr), str(nt)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["batch_device"]["frames_with_faces"] == out["batch_host"]["frames_with_faces"] > 0
    t = out["tracker"]
    assert t["vj_calls"] >= 2 and t["cs_calls"] >= 50 and t["after_60_calls"][4] == "CS"
    if "tracker_reference_js" in out:
        assert t["same_result_as_reference"] is True, (t["after_60_calls"], out["tracker_reference_js"]["after_60_calls"])


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_js_host_parity_on_gpu(tmp_path):
    _build()
    job = _parity_job(tmp_path)
    jf = tmp_path / "job.json"
    jf.write_text(json.dumps(job))
    # (with one visible GPU the sharded call has one rank and skips RCCL; tests/test_gpu_shapes.py::test_allgather_best_faces_over_rccl
    #  forces the RCCL path — ncclCommInitAll alone takes ~50 s on the test box, once is enough)
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "parity_gpu.js"), str(jf)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out["errors"]
    assert out["checked"] > 500
    exact, total = (int(v) for v in out["cs_parity"].split("/"))
    assert total > 0 and exact == total, f"camshift from JS: {out['cs_parity']} calls bit-exact"


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_js_host_benchmark_runs(tmp_path):
    """tests/js/bench_host.js (the js_host sub-record of bench.py): both detect paths and the drop-in tracker latency produce numbers,
    the pipelined DeviceBatch path is faster than the host-frames path, and the drop-in tracker ends where the reference JS ends."""
    import numpy as np

    _build()
    W, H, n, nt = 320, 240, 64, 30
    c2, tr = tmp_path / "c2.raw", tmp_path / "track.raw"
    synth.mixed_batch(n, W, H, seed0=1234).tofile(str(c2))
    np.stack([synth.face_frame(W, H, [(90 + 2 * k, 50 + k, 96)]) for k in range(nt)]).tofile(str(tr))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "bench_host.js"), "0.4", str(c2), str(n), str(tr), str(nt)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["batch_host"]["frames_per_s"] > 0 and out["batch_device"]["frames_per_s"] > out["batch_host"]["frames_per_s"]
    assert out["batch_device"]["frames_with_faces"] == out["batch_host"]["frames_with_faces"] > 0
    t = out["tracker"]
    assert t["vj_calls"] >= 4 and t["cs_calls"] >= 100 and t["after_60_calls"][4] == "CS"
    if "tracker_reference_js" in out:  # only where oracle/_ref was built (it travels with the snapshot)
        assert out["tracker"]["same_result_as_reference"] is True, (out["tracker"]["after_60_calls"], out["tracker_reference_js"]["after_60_calls"])
