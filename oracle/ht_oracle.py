"""ctypes binding of oracle/libht_oracle.so (the plain-C CPU restatement in oracle/ht_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
checker.  The product package (headtrackr_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libht_oracle.so")

MAX_LEVELS = 96


class Level(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("off", C.c_int64 * 4)]


HIT_DTYPE = np.dtype([("scale", "<i4"), ("q", "<i4"), ("x", "<i4"), ("y", "<i4"), ("sum", "<f8")])
RECT_DTYPE = np.dtype(
    [("x", "<f8"), ("y", "<f8"), ("width", "<f8"), ("height", "<f8"), ("confidence", "<f8"), ("neighbors", "<i4"), ("pad", "<i4")]
)


class CsState(C.Structure):
    _fields_ = [
        ("model", C.c_int32 * 4096),
        ("sw", C.c_int32 * 4),
        ("x", C.c_double),
        ("y", C.c_double),
        ("width", C.c_double),
        ("height", C.c_double),
        ("angle", C.c_double),
        ("calc_angles", C.c_int32),
        ("pad", C.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ht_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libht_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p = C.POINTER(C.c_uint8)
        L.ho_whitebalance.restype = C.c_double
        L.ho_whitebalance.argtypes = [u8p, C.c_int, C.c_int]
        L.ho_grayscale_rgba.argtypes = [u8p, C.c_int, C.c_int]
        L.ho_gray_plane.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.ho_scale.restype = C.c_double
        L.ho_scale.argtypes = [C.c_int]
        L.ho_pyramid.restype = C.c_int
        L.ho_pyramid.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Level), u8p, C.POINTER(C.c_int64)]
        L.ho_detect_raw.restype = C.c_int64
        L.ho_detect_raw.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_size_t, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.ho_hits_to_rects.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ho_group.restype = C.c_int
        L.ho_group.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ho_cs_init.argtypes = [C.POINTER(CsState), u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ho_cs_track.argtypes = [C.POINTER(CsState), u8p, C.c_int, C.c_int]
        L.ho_resample.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        assert L.ho_sizeof_level() == C.sizeof(Level)
        assert L.ho_sizeof_hit() == HIT_DTYPE.itemsize
        assert L.ho_sizeof_rect() == RECT_DTYPE.itemsize
        assert L.ho_sizeof_cs_state() == C.sizeof(CsState)
        _lib = L
    return _lib


def _u8(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def whitebalance(rgba: np.ndarray) -> float:
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    return float(lib().ho_whitebalance(_u8(rgba), w, h))


def grayscale_rgba(rgba: np.ndarray) -> np.ndarray:
    """ccv.grayscale: returns a gray RGBA copy."""
    out = np.ascontiguousarray(rgba, dtype=np.uint8).copy()
    h, w = out.shape[:2]
    lib().ho_grayscale_rgba(_u8(out), w, h)
    return out


def pyramid(rgba: np.ndarray, interval: int = 5, cw: int = 24, ch: int = 24, gray_in_r: bool = False):
    """Returns (levels, arena): levels = list of (w, h, [off0..off3]); arena = uint8 array with all planes."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    levels = (Level * MAX_LEVELS)()
    nbytes = C.c_int64(0)
    n = lib().ho_pyramid(_u8(rgba), w, h, int(gray_in_r), interval, cw, ch, levels, None, C.byref(nbytes))
    if n < 0:
        raise ValueError("pyramid layout failed")
    arena = np.zeros(nbytes.value + 16, dtype=np.uint8)
    lib().ho_pyramid(_u8(rgba), w, h, int(gray_in_r), interval, cw, ch, levels, _u8(arena), C.byref(nbytes))
    lv = [(levels[i].w, levels[i].h, [int(levels[i].off[s]) for s in range(4)]) for i in range(n)]
    return lv, arena[: nbytes.value]


def plane(levels, arena, i: int, slot: int = 0) -> np.ndarray:
    w, h, off = levels[i]
    return arena[off[slot] : off[slot] + w * h].reshape(h, w)


def detect_raw(rgba: np.ndarray, cascade_blob: bytes, interval: int = 5, gray_in_r: bool = False, cap: int = 1 << 16,
               stage_pass: np.ndarray | None = None) -> np.ndarray:
    """ccv.grayscale + ccv.detect_objects(..., min_neighbors=0): structured array of raw hits in emission order."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    blob = np.frombuffer(cascade_blob, dtype=np.uint8)
    hits = np.zeros(cap, dtype=HIT_DTYPE)
    sp = stage_pass.ctypes.data_as(C.POINTER(C.c_int64)) if stage_pass is not None else None
    n = lib().ho_detect_raw(_u8(rgba), w, h, int(gray_in_r), _u8(blob), blob.size, interval, hits.ctypes.data, cap, sp)
    if n < 0:
        raise ValueError("ho_detect_raw failed")
    if n > cap:
        raise OverflowError(f"{n} hits exceed cap {cap}")
    return hits[:n].copy()


def hits_to_rects(hits: np.ndarray, interval: int = 5, cw: int = 24, ch: int = 24) -> np.ndarray:
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    out = np.zeros(len(hits), dtype=RECT_DTYPE)
    if len(hits):
        lib().ho_hits_to_rects(hits.ctypes.data, len(hits), interval, cw, ch, out.ctypes.data)
    return out


def group(rects: np.ndarray, min_neighbors: int = 1) -> np.ndarray:
    rects = np.ascontiguousarray(rects, dtype=RECT_DTYPE)
    out = np.zeros(max(1, len(rects)), dtype=RECT_DTYPE)
    n = lib().ho_group(rects.ctypes.data, len(rects), min_neighbors, out.ctypes.data)
    return out[:n].copy()


def detect_objects(rgba, cascade_blob, interval=5, min_neighbors=1, gray_in_r=False):
    hits = detect_raw(rgba, cascade_blob, interval, gray_in_r)
    rects = hits_to_rects(hits, interval)
    return group(rects, min_neighbors) if min_neighbors > 0 else rects


class Camshift:
    """camshift.Tracker (camshift.js:148-354)."""

    def __init__(self, calc_angles: bool = True):
        self.s = CsState()
        self.calc_angles = calc_angles

    def init_tracker(self, rgba: np.ndarray, rect):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w = rgba.shape[:2]
        lib().ho_cs_init(C.byref(self.s), _u8(rgba), w, h, int(rect[0]), int(rect[1]), int(rect[2]), int(rect[3]), int(self.calc_angles))

    def track(self, rgba: np.ndarray):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w = rgba.shape[:2]
        lib().ho_cs_track(C.byref(self.s), _u8(rgba), w, h)
        return self.search_window(), self.track_obj()

    def search_window(self):
        return [int(v) for v in self.s.sw]

    def track_obj(self):
        return dict(x=self.s.x, y=self.s.y, width=self.s.width, height=self.s.height, angle=self.s.angle)


def cs_init(rgba: np.ndarray, x: int, y: int, w: int, h: int, calc_angles: bool = True) -> Camshift:
    t = Camshift(calc_angles)
    t.init_tracker(rgba, (x, y, w, h))
    return t


def cs_track(t: Camshift, rgba: np.ndarray):
    return t.track(rgba)


def cs_histograms(t: Camshift, rgba: np.ndarray):
    """(model histogram of the tracker, full-frame histogram of `rgba`) as camshift.Histogram computes them
    (camshift.js:49-72): 4096 bins each."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    px = rgba.reshape(-1, 4).astype(np.int64)
    bins = 256 * (px[:, 0] >> 4) + 16 * (px[:, 1] >> 4) + (px[:, 2] >> 4)
    return np.array(t.s.model, dtype=np.int64), np.bincount(bins, minlength=4096)


def best_faces(rgba_frames, cascade_blob, min_neighbors: int = 1) -> np.ndarray:
    """facetrackr.Tracker.doVJDetection's selection (facetrackr.js:147-175) per frame: grouped rect of highest confidence
    (strict '>', first wins); no detection -> zeros, confidence -10000, neighbors 0."""
    out = np.zeros(len(rgba_frames), dtype=RECT_DTYPE)
    for i, f in enumerate(rgba_frames):
        g = detect_objects(f, cascade_blob, 5, min_neighbors)
        out[i]["confidence"] = -10000.0
        for k in range(len(g)):
            if k == 0 or g[k]["confidence"] > out[i]["confidence"]:
                out[i] = g[k]
    return out
