"""Pins the CPU oracle (oracle/ht_oracle.c) to the reference: every golden vector in tests/golden/*.json was produced
by executing the UNMODIFIED reference JS (tests/golden/make_golden.py -> oracle/ref_harness.js).  Bit-exact for bytes,
indices and binary64 values; camshift `angle` (Math.atan2, libm-dependent) within 1e-12 rad."""
import math
import struct
import zlib

import numpy as np
import pytest

from conftest import load_golden
from headtrackr_amd import synth
from oracle import ht_oracle as ho

DETECT = load_golden("detect.json")
CAMSHIFT = load_golden("camshift.json")


def creation_order(nlevels, next_):
    """ccv.js:117-147: levels 1..n-1 (slot 0), then for i >= 2*next the variants 1,2,3."""
    return [(i, 0) for i in range(1, nlevels)] + [(i, s) for i in range(2 * next_, nlevels) for s in (1, 2, 3)]


def test_vote_template_pinned():
    assert synth.vote_template().tolist() == DETECT["vote_template"]


def test_v8_scale_constants():
    # the oracle hard-codes V8's Math.pow(2^(1/6), i); the golden file recorded what V8 produced
    assert ho.lib().ho_scale(5) == DETECT["scale6"]
    assert struct.pack("<d", DETECT["scale6_pows"][4]).hex() == "3e6e3da5fe65f93f"


@pytest.mark.parametrize("case", DETECT["cases"], ids=lambda c: c["name"])
def test_detect_case(case, cascade):
    w, h = case["w"], case["h"]
    interval = 5 if "interval3" not in case["name"] else 3
    frame = synth.make(case["gen"], w, h)
    assert zlib.crc32(frame.tobytes()) == case["input_crc"], "synthetic input drifted from the golden input"
    assert ho.whitebalance(frame) == case["whitebalance"]
    gray = ho.grayscale_rgba(frame)
    assert zlib.crc32(gray[..., 0].tobytes()) == case["gray_crc"]
    assert zlib.crc32(gray.tobytes()) == case["gray_rgba_crc"]

    levels, arena = ho.pyramid(frame, interval=interval)
    order = creation_order(len(levels), interval + 1)
    assert len(order) == len(case["pyramid"])
    for (i, s), g in zip(order, case["pyramid"]):
        p = ho.plane(levels, arena, i, s)
        assert (p.shape[1], p.shape[0]) == (g["w"], g["h"]), f"level {i} size"
        assert zlib.crc32(p.tobytes()) == g["crc"], f"level {i} slot {s} bytes"

    hits = ho.detect_raw(frame, cascade.blob, interval=interval)
    rects = ho.hits_to_rects(hits, interval=interval)
    assert len(rects) == len(case["raw"])
    for r, g in zip(rects, case["raw"]):
        for k in ("x", "y", "width", "height", "confidence"):
            assert r[k] == g[k], (k, r, g)
    # the same scan through the gray-in-R entry (what ccv.detect_objects itself sees)
    hits2 = ho.detect_raw(gray, cascade.blob, interval=interval, gray_in_r=True)
    assert hits2.tobytes() == hits.tobytes()

    grouped = ho.group(rects, case["min_neighbors"])
    assert len(grouped) == len(case["grouped"])
    for r, g in zip(grouped, case["grouped"]):
        for k in ("x", "y", "width", "height", "confidence", "neighbors"):
            assert r[k] == g[k], (k, r, g)


@pytest.mark.parametrize("case", CAMSHIFT["cases"], ids=lambda c: c["name"])
def test_camshift_case(case):
    w, h = case["w"], case["h"]
    frames = [synth.make(g, w, h) for g in case["gen"]]
    tr = ho.Camshift(calc_angles=case["calcAngles"])
    tr.init_tracker(frames[0], case["rect"])
    for call in case["calls"]:
        sw, to = tr.track(frames[call["frame"]])
        assert sw == call["sw"]
        for k in ("x", "y", "width", "height"):
            assert to[k] == call[k], (k, to, call)
        if call["angle"] is None:
            assert math.isnan(to["angle"])
        else:
            assert abs(to["angle"] - call["angle"]) <= 1e-12
