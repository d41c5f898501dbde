#!/usr/bin/env python3
"""Regenerates tests/golden/*.json by running the UNMODIFIED reference JS (needs /root/reference and node).

    python tests/golden/make_golden.py

Frames are synthesised by headtrackr_amd.synth (integer-only, reproducible), written as raw RGBA to a temp dir and
fed to oracle/ref_harness.js, which executes /root/reference/headtrackr.js on oracle/canvas_shim.js.  Each golden case
echoes its generator spec (`gen` / `gens`) so the tests rebuild exactly the same input and check its CRC.
This script is test infrastructure; it is the only place (with ref_harness.js) that touches the reference.
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from headtrackr_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def detect_cases():
    cs = []

    def add(name, w, h, gen, **kw):
        cs.append(dict(name=name, kind="detect", w=w, h=h, gen=gen, ops=["gray", "pyramid", "raw", "grouped", "whitebalance"], **kw))

    # C1: the reference's own CPU-runnable configuration, single 320x240 frame, full cascade
    add("c1_face_320x240", 320, 240, dict(family="face", faces=[[100, 60, 96]]))
    add("two_faces_320x240", 320, 240, dict(family="face", faces=[[20, 30, 48], [150, 90, 120]]))
    add("noise_320x240", 320, 240, dict(family="noise", seed=1234))
    add("smooth_320x240", 320, 240, dict(family="smooth", seed=1235))
    for i in range(6):
        add(f"mixed{i}_320x240", 320, 240, dict(family="mixed", index=i, seed0=1234))
    add("face_160x120", 160, 120, dict(family="face", faces=[[40, 20, 72]]))
    add("noise_160x120", 160, 120, dict(family="noise", seed=99))
    add("face_odd_201x157", 201, 157, dict(family="face", faces=[[50, 30, 90]]))
    add("smooth_odd_203x151", 203, 151, dict(family="smooth", seed=5))
    add("face_tiny_100x90", 100, 90, dict(family="face", faces=[[10, 8, 70]]))
    add("face_empty_levels_64x48", 64, 48, dict(family="face", faces=[[8, 4, 40]]))
    add("face_minneighbors2_320x240", 320, 240, dict(family="face", faces=[[100, 60, 96]]), min_neighbors=2)
    add("face_interval3_160x120", 160, 120, dict(family="face", faces=[[40, 20, 72]]), interval=3)
    add("faces_1280x720", 1280, 720, dict(family="face", faces=[[400, 200, 240], [900, 100, 64], [100, 500, 150]]))
    add("smooth_1280x720", 1280, 720, dict(family="smooth", seed=77))
    return cs


def camshift_cases():
    cs = []
    W, H = 320, 240
    # static vote-face target, tracked 4x on the same frame (converges)
    cs.append(dict(name="cs_static_face", kind="camshift", w=W, h=H, calcAngles=True, rect=[102, 62, 90, 90], repeat=4,
                   gens=[dict(family="face", faces=[[100, 60, 96]])] * 2))
    # anisotropic rotated blob walking <= 3 px / step
    walk = synth.lcg_stream(4242, 40).astype("int64") >> 20
    gens, cx, cy = [], 150, 110
    for k in range(14):
        gens.append(dict(family="blob", cx=cx, cy=cy, a=44, b=20, rot=[4, 3, 5], color=[200, 60, 40], seed=7 + k))
        cx += int(walk[2 * k] % 7) - 3
        cy += int(walk[2 * k + 1] % 7) - 3
    cs.append(dict(name="cs_walk_blob", kind="camshift", w=W, h=H, calcAngles=True, rect=[110, 80, 80, 60], gens=gens))
    cs.append(dict(name="cs_walk_blob_noangles", kind="camshift", w=W, h=H, calcAngles=False, rect=[110, 80, 80, 60], gens=gens))
    # model = blob colours only (rect inside the blob); target vanishes after 3 frames -> zero mass -> 0x0 ("lost", SURVEY.md §5 failure detection)
    gens2 = [dict(family="blob", cx=100, cy=100, a=30, b=24, color=[30, 220, 60], seed=3, bg="flat")] * 3 + \
            [dict(family="face", faces=[], gray=110)] * 3
    cs.append(dict(name="cs_lost", kind="camshift", w=W, h=H, calcAngles=True, rect=[85, 90, 30, 20], gens=gens2))
    # target at the border: window clamping / shifting (camshift.js:286-289), init rect partly outside the canvas
    gens3 = [dict(family="blob", cx=300 + k, cy=20 - k, a=26, b=18, rot=[3, 4, 5], color=[40, 80, 230], seed=11 + k) for k in range(8)]
    cs.append(dict(name="cs_border", kind="camshift", w=W, h=H, calcAngles=True, rect=[280, -6, 60, 50], gens=gens3))
    # 720p
    gens4 = [dict(family="blob", cx=640 + 3 * k, cy=360 - 2 * k, a=120, b=60, rot=[12, 5, 13], color=[210, 140, 40], seed=21 + k) for k in range(5)]
    cs.append(dict(name="cs_720p", kind="camshift", w=1280, h=720, calcAngles=True, rect=[520, 290, 240, 140], gens=gens4))
    return cs


def facetrackr_cases():
    cs = []
    W, H = 320, 240
    face = dict(family="face", faces=[[100, 60, 96]])
    # default params: 15 x "WB", then "VJ", then "CS" 
    cs.append(dict(name="ft_default_static", kind="facetrackr", w=W, h=H, params={}, gens=[face] * 20))
    # no whitebalancing, angles on, face drifting 2 px / frame
    gens = [dict(family="face", faces=[[100 + 2 * k, 60 + k, 96]]) for k in range(8)]
    cs.append(dict(name="ft_nowb_moving", kind="facetrackr", w=W, h=H, params=dict(whitebalancing=False, calcAngles=True), gens=gens))
    # nothing to find: stays in "VJ" with confidence -10000
    cs.append(dict(name="ft_noface", kind="facetrackr", w=W, h=H, params=dict(whitebalancing=False),
                   gens=[dict(family="noise", seed=50 + k) for k in range(3)]))
    return cs


def post_cases():
    """Smoother, headposition and the per-frame body of headtrackr.Tracker (host post-processing)."""
    cs = []
    r = synth.lcg_stream(777, 400).astype("int64") >> 16
    pos = [[100 + int(r[5 * i] % 40), 80 + int(r[5 * i + 1] % 30), 50 + int(r[5 * i + 2] % 9), 90 + int(r[5 * i + 3] % 12), 100 + int(r[5 * i + 4] % 12)] for i in range(25)]
    cs.append(dict(name="smoother_default", kind="smoother", alpha=0.35, interval=35, positions=pos))
    cs.append(dict(name="smoother_late_init", kind="smoother", alpha=0.5, interval=20, positions=pos[:8], init_at=3))
    faces = [[160 + int(r[100 + 4 * i] % 200) - 100, 120 + int(r[101 + 4 * i] % 160) - 80, 80 + int(r[102 + 4 * i] % 30), 96 + int(r[103 + 4 * i] % 30)] for i in range(30)]
    faces += [[30, 120, 80, 100], [300, 30, 70, 90], [20, 20, 60, 70], [160, 230, 90, 110], [310, 235, 50, 60]]  # edge / corner corrections
    cs.append(dict(name="headposition_default", kind="headposition", camw=320, camh=240, params={}, faces=faces))
    cs.append(dict(name="headposition_fov_noedge", kind="headposition", camw=320, camh=240, faces=faces,
                   params=dict(fov=45, edgecorrection=False, distance_from_camera_to_screen=8)))
    W, H = 320, 240
    gens = [dict(family="face", faces=[[100 + k, 60 + (k % 3), 96]]) for k in range(14)] + \
           [dict(family="face", faces=[], gray=250)] * 2 + [dict(family="face", faces=[[60, 50, 110]])] * 10
    cs.append(dict(name="pipeline_nowb", kind="pipeline", w=W, h=H, whitebalancing=False, params=dict(calcAngles=True), gens=gens))
    return cs


def mainjs_cases():
    """The reference's own headtrackr.Tracker loop (main.js) with a debug canvas: whitebalance phase, VJ box, rotated CS boxes,
    a lost track and the re-detection."""
    W, H = 320, 240
    gens = [dict(family="face", faces=[[100, 60, 96]])] * 16 + [dict(family="face", faces=[[100 + 2 * k, 60 + k, 96]]) for k in range(1, 9)] + \
           [dict(family="face", faces=[], gray=250)] * 2 + [dict(family="face", faces=[[60, 50, 110]])] * 4
    return [dict(name="mainjs_debug_angles", kind="mainjs", w=W, h=H, debug=True, params=dict(calcAngles=True), gens=gens),
            dict(name="mainjs_debug_noangles", kind="mainjs", w=W, h=H, debug=True, params=dict(calcAngles=False, smoothing=False), gens=gens[:22])]


def large_cases():
    """The largest frame size of BASELINE.json (configs[4]: 1920x1080 feeds): pins the ORACLE to the reference there.  A file of its own,
    read only by the CPU test of the oracle (tests/test_oracle_golden.py) — the GPU suite compares the HIP path with the oracle at this
    size (pyramid planes, the C5 loop), so reference -> oracle -> GPU holds at 1080p as well."""
    W, H = 1920, 1080
    cs = []
    ops = ["gray", "pyramid", "raw", "grouped", "whitebalance"]
    cs.append(dict(name="faces_1920x1080", kind="detect", w=W, h=H, ops=ops,
                   gen=dict(family="face", faces=[[700, 300, 400], [1500, 120, 90], [200, 700, 260]])))
    cs.append(dict(name="smooth_1920x1080", kind="detect", w=W, h=H, ops=ops, gen=dict(family="smooth", seed=1080)))
    # C5's camshift shape: a ~360 x 360 search window on a 1080p frame, target walking 4 px / frame
    gens = [dict(family="blob", cx=960 + 4 * k, cy=540 - 3 * k, a=180, b=110, rot=[12, 5, 13], color=[210, 140, 40], seed=31 + k) for k in range(5)]
    cs.append(dict(name="cs_1080p", kind="camshift", w=W, h=H, calcAngles=True, rect=[780, 400, 360, 280], gens=gens))
    return cs


def run(cases, out_name):
    with tempfile.TemporaryDirectory() as td:
        cache = {}

        def frame_file(gen, w, h):
            key = json.dumps([gen, w, h], sort_keys=True)
            if key not in cache:
                fn = f"f{len(cache)}.raw"
                synth.make(gen, w, h).tofile(os.path.join(td, fn))
                cache[key] = fn
            return cache[key]

        job = {"cases": []}
        for c in cases:
            c = dict(c)
            if "gen" in c:
                c["frame"] = frame_file(c["gen"], c["w"], c["h"])
            if "gens" in c:
                c["frames"] = [frame_file(g, c["w"], c["h"]) for g in c["gens"]]
                c["gen"] = c["gens"]
            if c["kind"] in ("smoother", "headposition"):
                c["gen"] = None
            job["cases"].append(c)
        jf = os.path.join(td, "job.json")
        with open(jf, "w") as f:
            json.dump(job, f)
        of = os.path.join(td, "out.json")
        subprocess.check_call(["node", os.path.join(ROOT, "oracle", "ref_harness.js"), jf, of])
        with open(of) as f:
            res = json.load(f)
    res["vote_template"] = synth.vote_template().tolist()
    with open(os.path.join(OUT, out_name), "w") as f:
        json.dump(res, f, separators=(",", ":"))
    print("wrote", out_name, os.path.getsize(os.path.join(OUT, out_name)), "bytes")


if __name__ == "__main__":
    run(detect_cases(), "detect.json")
    run(camshift_cases(), "camshift.json")
    run(facetrackr_cases(), "facetrackr.json")
    run(post_cases(), "post.json")
    run(mainjs_cases(), "debug.json")
    run(large_cases(), "large.json")
