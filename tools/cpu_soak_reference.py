#!/usr/bin/env python3
"""tools/cpu_soak_reference.py [seconds] [seed] — randomised soak of the ORACLE against the REFERENCE itself, on the CPU.

tests/golden/*.json pin oracle/ht_oracle.c to the unmodified reference JS on a fixed list of cases; tools/gpu_soak.py compares the HIP
path with the oracle on thousands of random geometries.  This closes the remaining link: random geometries (24 ... 1400 px), frame
families, intervals, min_neighbors and tracker set-ups are run through /root/reference/headtrackr.js (oracle/ref_harness.js on
oracle/canvas_shim.js, exactly as tests/golden/make_golden.py does) and through the oracle, and compared with the SAME checks as
tests/test_oracle_golden.py: input CRC, whitebalance, gray bytes, every pyramid plane (size + CRC), raw rects incl. the binary64
confidence, grouped rects; camshift search window / x / y / width / height exact and the angle to 1e-12.
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


def main():
    t0 = time.time()
    tot = dict(detect=0, planes=0, raw=0, grouped=0, camshift=0, calls=0, sizes=set())
    batch_no = 0
    with tempfile.TemporaryDirectory() as td:
        mg.OUT = td
        while time.time() - t0 < BUDGET:
            cases = [detect_case(batch_no * 100 + i) for i in range(16)] + [camshift_case(batch_no * 100 + i) for i in range(8)]
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
            for spec, got in zip(cases, res["cases"]):
                try:
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
          f"{tot['grouped']} grouped faces, {tot['calls']} track() calls (window / x / y / width / height exact, angle to 1e-12): all identical")
    return 0


if __name__ == "__main__":
    sys.exit(main())
