#!/usr/bin/env python3
"""tools/cpu_soak_reference.py [seconds] [seed] — randomised soak of the ORACLE against the REFERENCE itself, on the CPU.

tests/golden/*.json pin oracle/ht_oracle.c to the unmodified reference JS on a fixed list of cases; tools/gpu_soak.py compares the HIP
path with the oracle on thousands of random geometries.  This closes the remaining link: random geometries (24 ... 1400 px), frame
families, intervals, min_neighbors and tracker set-ups are run through /root/reference/headtrackr.js (oracle/ref_harness.js on
oracle/canvas_shim.js, exactly as tests/golden/make_golden.py does) and through the oracle, and compared with the SAME checks as
tests/test_oracle_golden.py: input CRC, whitebalance, gray bytes, every pyramid plane (size + CRC), raw rects incl. the binary64
confidence, grouped rects; camshift search window / x / y / width / height exact and the angle to 1e-12.  On the same reference
output the PRODUCT's host-side code is checked too (no GPU needed): ht_group_rects of libheadtrackr_hip.so, and through
tests/js/facade_cpu.js the JS facade's grouping, Smoother and headposition.Tracker on random sequences.
Needs /root/reference and node (this container, not the GPU box).  Exit 1 on the first mismatch, with the case's generator spec."""
import importlib.util
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

from headtrackr_amd.cascade import load_cascade  # noqa: E402


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) & 0x7FFFFFFF
rng = np.random.default_rng(SEED)
mg = load(os.path.join(ROOT, "tests", "golden", "make_golden.py"), "make_golden")
tg = load(os.path.join(ROOT, "tests", "test_oracle_golden.py"), "test_oracle_golden")
cascade = load_cascade()
# the JS facade's state machines need the oracle as a Node addon (tests/test_js_host.py builds it: tests/js/oracle_addon.node)
MOCK = os.path.exists(os.path.join(ROOT, "tests", "js", "oracle_addon.node"))


def rand_gen(w, h):
    k = int(rng.integers(0, 4))
    seed = int(rng.integers(1, 1 << 30))
    if k == 0:
        return dict(family="noise", seed=seed)
    if k == 1:
        return dict(family="smooth", seed=seed)
    faces = []
    for _ in range(int(rng.integers(1, 4))):
        s = int(rng.integers(24, max(25, min(w, h) + 1)))
        if s <= min(w, h):
            faces.append([int(rng.integers(0, w - s + 1)), int(rng.integers(0, h - s + 1)), s])
    return dict(family="face", faces=faces, gray=int(rng.integers(60, 180)))


def detect_case(i):
    big = rng.random() < 0.12
    w = int(rng.integers(24, 1400 if big else 520))
    h = int(rng.integers(24, 900 if big else 400))
    c = dict(name=f"soak{i}", kind="detect", w=w, h=h, gen=rand_gen(w, h), ops=["gray", "pyramid", "raw", "grouped", "whitebalance"],
             min_neighbors=int(rng.integers(1, 4)))
    if rng.random() < 0.2:
        c.update(name=f"soak{i}_interval3", interval=3)
    return c


def camshift_case(i):
    big = rng.random() < 0.2
    w = int(rng.integers(64, 1300 if big else 480))
    h = int(rng.integers(48, 800 if big else 360))
    a, b = int(rng.integers(6, max(7, w // 4))), int(rng.integers(5, max(6, h // 4)))
    cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
    color = [int(v) for v in rng.integers(20, 236, 3)]
    rot = [int(rng.integers(1, 14)), int(rng.integers(0, 9)), int(rng.integers(1, 14))]
    gens = []
    for k in range(int(rng.integers(3, 9))):
        gens.append(dict(family="blob", cx=cx, cy=cy, a=a, b=b, rot=rot, color=color, seed=int(rng.integers(1, 1 << 20)),
                         bg="flat" if rng.random() < 0.3 else "noise"))
        cx += int(rng.integers(-3, 4))
        cy += int(rng.integers(-3, 4))
    if rng.random() < 0.15:  # the target vanishes: zero mass, 0 x 0 track object (camshift.js:222-259 with NaN sums)
        gens += [dict(family="face", faces=[], gray=110)] * 2
    rw, rh = int(rng.integers(4, max(5, 2 * a + 8))), int(rng.integers(4, max(5, 2 * b + 8)))
    rect = [gens[0]["cx"] - rw // 2 + int(rng.integers(-6, 7)), gens[0]["cy"] - rh // 2 + int(rng.integers(-6, 7)), rw, rh]  # may leave the canvas
    return dict(name=f"soakcs{i}", kind="camshift", w=w, h=h, calcAngles=bool(rng.random() < 0.7), rect=rect, gens=gens)


def post_cases(i):
    """headtrackr.Smoother (smoother.js:13-88) and headposition.Tracker (headposition.js:35-201): pure host arithmetic of SURVEY §8(f3),
    restated in headtrackr_amd/js/tracker.js"""
    n = int(rng.integers(3, 30))
    pos = [[float(v) for v in rng.uniform(-50, 700, 5)] for _ in range(n)]
    if rng.random() < 0.5:
        pos = [[int(v) for v in p] for p in pos]
    sm = dict(name=f"soaksm{i}", kind="smoother", alpha=float(rng.choice([0.35, 0.5, 0.1, 0.9, float(rng.uniform(0.01, 0.99))])),
              interval=int(rng.integers(1, 60)), positions=pos)
    if rng.random() < 0.3:
        sm["init_at"] = int(rng.integers(0, n))
    camw, camh = int(rng.integers(64, 1921)), int(rng.integers(48, 1081))
    faces = [[float(rng.uniform(-20, camw + 20)), float(rng.uniform(-20, camh + 20)), float(rng.uniform(8, camw / 2)), float(rng.uniform(8, camh / 2))]
             for _ in range(int(rng.integers(2, 30)))]
    params = {}
    if rng.random() < 0.5:
        params["fov"] = float(rng.uniform(20, 90))
    if rng.random() < 0.5:
        params["edgecorrection"] = bool(rng.random() < 0.5)
    if rng.random() < 0.5:
        params["distance_from_camera_to_screen"] = float(rng.uniform(0, 30))
    if rng.random() < 0.3:
        params["distance_to_screen"] = float(rng.uniform(30, 120))
    return [sm, dict(name=f"soakhp{i}", kind="headposition", camw=camw, camh=camh, params=params, faces=faces)]


def face_walk(w, h, n, vanish_at=None):
    """gens of n frames: one vote-image face walking <= 3 px per frame; from vanish_at on a flat gray frame (the track is lost)"""
    s0 = int(rng.integers(40, max(41, min(w, h) - 16)))
    x, y = int(rng.integers(0, w - s0 + 1)), int(rng.integers(0, h - s0 + 1))
    gens = []
    for k in range(n):
        if vanish_at is not None and k >= vanish_at:
            gens.append(dict(family="face", faces=[], gray=250))
        else:
            gens.append(dict(family="face", faces=[[x, y, s0]]))
        x = int(np.clip(x + rng.integers(-3, 4), 0, w - s0))
        y = int(np.clip(y + rng.integers(-3, 4), 0, h - s0))
    return gens


def facade_cases(i):
    """the reference's stateful objects on random sequences: facetrackr.Tracker (facetrackr.js:37-228: WB -> VJ -> CS), the per-frame body of
    headtrackr.Tracker composed from them (main.js:168-305: lost track, Smoother, headposition) and the reference's own main.js loop with a
    debug canvas (main.js:199-219).  Compared through tests/js/parity_cpu.js: the product's facade on the oracle-backed mock addon."""
    out = []
    w, h = [(160, 120), (200, 150), (320, 240), (int(rng.integers(96, 400)), int(rng.integers(72, 300)))][int(rng.integers(0, 4))]
    wb = bool(rng.random() < 0.4)
    n = int(rng.integers(18, 23)) if wb else int(rng.integers(5, 12))
    fam = rng.random()
    if fam < 0.15:
        gens = [dict(family="noise", seed=int(rng.integers(1, 1 << 20))) for _ in range(n)]  # nothing to find
    else:
        gens = face_walk(w, h, n, vanish_at=(int(rng.integers(n - 4, n)) if fam < 0.4 and n > 6 else None))
    out.append(dict(name=f"soakft{i}", kind="facetrackr", w=w, h=h, gens=gens,
                    params=dict(whitebalancing=wb, calcAngles=bool(rng.random() < 0.5))))
    n = int(rng.integers(8, 20))
    gens = face_walk(w, h, n, vanish_at=(int(rng.integers(4, n - 2)) if rng.random() < 0.5 else None))
    if rng.random() < 0.5:
        gens += face_walk(w, h, int(rng.integers(3, 8)))  # a face again: re-detection
    out.append(dict(name=f"soakpl{i}", kind="pipeline", w=w, h=h, whitebalancing=False, gens=gens,
                    params=dict(calcAngles=bool(rng.random() < 0.5))))
    if i % 4 == 0:  # the main.js loop starts with its whitebalance phase: 16 identical frames first
        first = face_walk(320, 240, 1)
        gens = first * 16 + face_walk(320, 240, int(rng.integers(4, 10)), vanish_at=(3 if rng.random() < 0.5 else None))
        out.append(dict(name=f"soakmj{i}", kind="mainjs", w=320, h=240, debug=True, gens=gens,
                        params=dict(calcAngles=bool(rng.random() < 0.5), smoothing=bool(rng.random() < 0.7))))
    return out


def check_facade(td, specs, res):
    """facetrackr / pipeline / mainjs results of the reference -> a job for tests/js/parity_cpu.js (the unchanged product facade on the mock addon)"""
    import subprocess

    from headtrackr_amd import synth

    fdir = os.path.join(td, "frames")
    os.makedirs(fdir, exist_ok=True)
    cache = {}

    def ffile(gen, w, h):
        key = json.dumps([gen, w, h], sort_keys=True)
        if key not in cache:
            cache[key] = os.path.join("frames", f"f{len(cache)}.raw")
            synth.make(gen, w, h).tofile(os.path.join(td, cache[key]))
        return cache[key]

    job = {"detect": [], "camshift": [], "facetrackr": [], "pipeline": [], "mainjs": [], "cpu_mock": True}
    for spec, c in zip(specs, res["cases"]):
        fr = None if c["kind"] not in ("facetrackr", "pipeline", "mainjs") else [ffile(g, c["w"], c["h"]) for g in spec["gens"]]
        if c["kind"] == "facetrackr":
            job["facetrackr"].append(dict(name=c["name"], w=c["w"], h=c["h"], frames=fr, golden={k: c[k] for k in ("params", "calls", "events")}))
        elif c["kind"] == "pipeline":
            g = {k: c[k] for k in ("params", "calls", "fov")}
            g["whitebalancing"] = spec.get("whitebalancing", True)
            job["pipeline"].append(dict(name=c["name"], w=c["w"], h=c["h"], frames=fr, golden=g))
        elif c["kind"] == "mainjs":
            job["mainjs"].append(dict(name=c["name"], w=c["w"], h=c["h"], frames=fr, golden={k: c[k] for k in ("params", "calls", "fov")}))
    with open(os.path.join(td, "facade_job.json"), "w") as f:
        json.dump(job, f)
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "parity_cpu.js"), os.path.join(td, "facade_job.json")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["ok"], ("JS facade (state machines) vs reference", j["errors"][:6])
    for f in cache.values():
        os.remove(os.path.join(td, f))
    return j["checked"], sum(len(c["frames"]) for k in ("facetrackr", "pipeline", "mainjs") for c in job[k])


def lib_group(rects, min_neighbors):
    """the product's own host grouping (ht_group_rects, csrc/ht_hostpost.h: what the Python / C hosts and ht_detect_collect_best use)"""
    import ctypes as C

    from headtrackr_amd import native

    out = np.zeros(max(1, len(rects)), dtype=native.RECT_DTYPE)
    n = C.c_uint32(0)
    assert native.lib().ht_group_rects(rects.ctypes.data, len(rects), min_neighbors, out.ctypes.data, C.byref(n)) == 0
    return out[: n.value]


def check_product_host(td, res):
    """The PRODUCT's host code against the reference's output of the same random cases (no GPU involved): ht_group_rects in C, and —
    through tests/js/facade_cpu.js — the JS facade's grouping (ccv._group), Smoother and headposition.Tracker."""
    import subprocess

    from headtrackr_amd import native

    det = [c for c in res["cases"] if c["kind"] == "detect"]
    for c in det:
        raw = np.zeros(len(c["raw"]), dtype=native.RECT_DTYPE)
        for k in ("x", "y", "width", "height", "confidence"):
            raw[k] = [r[k] for r in c["raw"]]
        raw["neighbors"] = 1
        got = lib_group(raw, c["min_neighbors"])
        assert len(got) == len(c["grouped"]), (c["name"], "ht_group_rects count")
        for g, w in zip(got, c["grouped"]):
            for k in ("x", "y", "width", "height", "confidence", "neighbors"):
                assert g[k] == w[k], (c["name"], "ht_group_rects", k, g, w)
    from headtrackr_amd import synth
    from oracle import ht_oracle as ho

    for c in det:  # raw hits in the addon's index form (from the oracle: the checker) for the facade's own seq construction
        iv = 3 if "interval3" in c["name"] else 5
        hits = ho.detect_raw(synth.make(c["gen"], c["w"], c["h"]), cascade.blob, interval=iv)
        c["interval"] = iv
        c["hits"] = {k: [float(v) if k == "sum" else int(v) for v in hits[k]] for k in ("scale", "q", "x", "y", "sum")}
    with open(os.path.join(td, "soak_detect.json"), "w") as f:
        json.dump(dict(cases=det), f)
    with open(os.path.join(td, "soak_post.json"), "w") as f:
        json.dump(dict(cases=[c for c in res["cases"] if c["kind"] in ("smoother", "headposition")]), f)
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "facade_cpu.js"), os.path.join(td, "soak_detect.json"),
                        os.path.join(td, "soak_post.json")], capture_output=True, text=True, timeout=300)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["ok"], ("JS facade vs reference", j["errors"][:5])


def main():
    t0 = time.time()
    tot = dict(detect=0, planes=0, raw=0, grouped=0, camshift=0, calls=0, post=0, facade_checks=0, facade_frames=0, sizes=set())
    batch_no = 0
    with tempfile.TemporaryDirectory() as td:
        mg.OUT = td
        while time.time() - t0 < BUDGET:
            cases = [detect_case(batch_no * 100 + i) for i in range(16)] + [camshift_case(batch_no * 100 + i) for i in range(8)]
            for i in range(4):
                cases += post_cases(batch_no * 100 + i)
            if MOCK:
                for i in range(2):
                    cases += facade_cases(batch_no * 100 + i)
            sys.stdout.flush()
            sys.stderr.flush()
            keep, null = (os.dup(1), os.dup(2)), os.open(os.devnull, os.O_WRONLY)
            os.dup2(null, 1)  # the harness (a child process) prints a line per case
            os.dup2(null, 2)
            try:
                mg.run(cases, "soak.json")
            finally:
                sys.stdout.flush()
                os.dup2(keep[0], 1)
                os.dup2(keep[1], 2)
                for fd in keep + (null,):
                    os.close(fd)
            with open(os.path.join(td, "soak.json")) as f:
                res = json.load(f)
            try:
                check_product_host(td, res)
                tot["post"] += 8
                if MOCK:
                    chk, frames = check_facade(td, cases, res)
                    tot["facade_checks"] += chk
                    tot["facade_frames"] += frames
            except AssertionError as e:
                print("MISMATCH product host code vs reference, batch", batch_no, "seed", SEED, "\n", e)
                return 1
            for spec, got in zip(cases, res["cases"]):
                try:
                    if got["kind"] in ("smoother", "headposition", "facetrackr", "pipeline", "mainjs"):
                        continue
                    if got["kind"] == "detect":
                        tg.check_detect_case(got, cascade)
                        tot["detect"] += 1
                        tot["planes"] += len(got["pyramid"])
                        tot["raw"] += len(got["raw"])
                        tot["grouped"] += len(got["grouped"])
                    else:
                        tg.check_camshift_case(got)
                        tot["camshift"] += 1
                        tot["calls"] += len(got["calls"])
                    tot["sizes"].add((got["w"], got["h"]))
                except AssertionError as e:
                    print("MISMATCH oracle vs reference:", json.dumps(spec), "\n", e)
                    return 1
            batch_no += 1
    print(f"oracle vs reference JS soak, seed {SEED}, {time.time() - t0:.0f} s: {tot['detect']} detect cases + {tot['camshift']} camshift set-ups over "
          f"{len(tot['sizes'])} geometries; {tot['planes']} pyramid planes (size + CRC), {tot['raw']} raw rects incl. confidence bits, "
          f"{tot['grouped']} grouped faces, {tot['calls']} track() calls (window / x / y / width / height exact, angle to 1e-12): all identical.  "
          f"Product host code on the same reference output: ht_group_rects (C) and the JS facade's seq construction + ccv._group on every raw list, "
          f"{tot['post']} random Smoother / headposition sequences through headtrackr_amd/js: all identical"
          + (f"; facetrackr.Tracker / headtrackr.Tracker body / main.js loop on {tot['facade_frames']} frames of random sequences through the "
             f"unchanged facade on the oracle-backed mock addon (tests/js/parity_cpu.js): {tot['facade_checks']} checks, all passed" if MOCK else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
