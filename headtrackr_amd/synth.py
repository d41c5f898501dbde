"""Seeded synthetic RGBA frames for parity tests and benchmarks.

Everything here is integer arithmetic (plus one fixed-order float64 accumulation for the vote template,
whose u8 result is pinned in tests/golden) so that the same frames come out on every machine: the golden
fixtures under tests/golden were produced by feeding exactly these frames to the reference JS.

Families:  N = LCG noise (0 detections, shortest cascade paths), S = smooth waves + light noise
(deep-stage survivors), F = flat gray frame with "vote image" faces derived from the cascade itself
(real detections), all RGBA u8 with A = 255.
"""
from __future__ import annotations

import functools

import numpy as np

from .cascade import Cascade, load_cascade

_LCG_A = 1664525
_LCG_C = 1013904223
_M32 = 0xFFFFFFFF


def lcg_stream(seed: int, n: int) -> np.ndarray:
    """First n outputs x1..xn of x <- x*1664525 + 1013904223 (mod 2^32), x0 = seed; returned as uint32."""
    if n <= 0:
        return np.zeros(0, dtype=np.uint32)
    # x_k = A_k * x0 + C_k ; build (A_k, C_k) for k = 1..n by doubling
    A = np.empty(n, dtype=np.uint64)
    C = np.empty(n, dtype=np.uint64)
    A[0] = _LCG_A
    C[0] = _LCG_C
    filled = 1
    while filled < n:
        m = min(filled, n - filled)
        ab = A[filled - 1]  # A_filled, C_filled : advance by `filled` steps
        cb = C[filled - 1]
        # x_{k+f} = A_k*(A_f*x0 + C_f) + C_k  ->  A_{k+f} = A_k*A_f ; C_{k+f} = A_k*C_f + C_k
        A[filled : filled + m] = (A[:m] * ab) & _M32
        C[filled : filled + m] = (A[:m] * cb + C[:m]) & _M32
        filled += m
    return ((A * np.uint64(seed & _M32) + C) & _M32).astype(np.uint32)


def noise_frame(w: int, h: int, seed: int) -> np.ndarray:
    """Family N: R,G,B = successive (lcg >> 24); A = 255."""
    s = lcg_stream(seed, 3 * w * h)
    rgb = (s >> np.uint32(24)).astype(np.uint8).reshape(h, w, 3)
    out = np.empty((h, w, 4), dtype=np.uint8)
    out[..., :3] = rgb
    out[..., 3] = 255
    return out


def _isin(x: np.ndarray, period: int, amp: int) -> np.ndarray:
    """Integer pseudo-sine: parabolic arches of the given period and amplitude (floor arithmetic)."""
    half = period // 2
    p = np.mod(x, 2 * half)
    neg = p >= half
    q = np.where(neg, p - half, p).astype(np.int64)
    s = (4 * amp * q * (half - q)) // (half * half)
    return np.where(neg, -s, s)


def smooth_frame(w: int, h: int, seed: int) -> np.ndarray:
    """Family S: v = 128 + wave(x) + wave(y) + (noise-128)//8 ; (R,G,B) = (v, 4v//5, 3v//5)."""
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    nz = (lcg_stream(seed, w * h) >> np.uint32(24)).astype(np.int64).reshape(h, w)
    v = 128 + _isin(x + 7 * (seed % 13), 106, 60) + _isin(y + 5 * (seed % 7), 144, 50) + (nz - 128) // 8
    v = np.clip(v, 0, 255)
    out = np.empty((h, w, 4), dtype=np.uint8)
    out[..., 0] = v
    out[..., 1] = (4 * v) // 5
    out[..., 2] = (3 * v) // 5
    out[..., 3] = 255
    return out


@functools.lru_cache(maxsize=4)
def _vote_template_cached(path: str | None) -> bytes:
    return _vote_template(load_cascade(path)).tobytes()


def _vote_template(c: Cascade) -> np.ndarray:
    """24x24 u8 'vote image': every feature adds its gain to its positive points and subtracts it from its
    negative points (a plane-z point covers a 2^z x 2^z block, gain / 4^z per pixel); normalised to [30,230]."""
    W, H = c.width, c.height
    V = np.zeros((H, W), dtype=np.float64)
    for f in c.features:
        gain = float(f["alpha"][1]) - float(f["alpha"][0])
        for q in range(int(f["size"])):
            for xs, ys, zs, sgn in ((f["px"], f["py"], f["pz"], 1.0), (f["nx"], f["ny"], f["nz"], -1.0)):
                z = int(zs[q])
                if z < 0:
                    continue
                b = 1 << z
                x0, y0 = int(xs[q]) * b, int(ys[q]) * b
                V[y0 : y0 + b, x0 : x0 + b] += sgn * gain / float(b * b)
    lo, hi = V.min(), V.max()
    T = np.floor(30.0 + (V - lo) * (200.0 / (hi - lo)) + 0.5)
    return np.clip(T, 0, 255).astype(np.uint8)


def vote_template(cascade_path: str | None = None) -> np.ndarray:
    return np.frombuffer(_vote_template_cached(cascade_path), dtype=np.uint8).reshape(24, 24).copy()


def _upscale_fixed(T: np.ndarray, s: int) -> np.ndarray:
    """Centre-aligned bilinear upscale of a square u8 image to s x s in 16.16 fixed point."""
    n = T.shape[0]
    i = np.arange(s, dtype=np.int64)
    pos = ((2 * i + 1) * n - s) * 65536 // (2 * s)
    pos = np.clip(pos, 0, (n - 1) << 16)
    i0 = pos >> 16
    t = pos & 0xFFFF
    i1 = np.minimum(i0 + 1, n - 1)
    Ti = T.astype(np.int64)
    rows = (Ti[i0, :] * (65536 - t)[:, None] + Ti[i1, :] * t[:, None] + 32768) >> 16  # s x n
    out = (rows[:, i0] * (65536 - t)[None, :] + rows[:, i1] * t[None, :] + 32768) >> 16  # s x s
    return out.astype(np.int64)


def paste_face(frame: np.ndarray, x: int, y: int, s: int, cascade_path: str | None = None) -> None:
    """Paste an s x s vote-image face (tint R=v, G=3v>>2, B=3v//5) with top-left (x, y) into an RGBA frame."""
    v = _upscale_fixed(vote_template(cascade_path), s)
    h, w = frame.shape[:2]
    x0, y0, x1, y1 = max(x, 0), max(y, 0), min(x + s, w), min(y + s, h)
    if x1 <= x0 or y1 <= y0:
        return
    sub = v[y0 - y : y1 - y, x0 - x : x1 - x]
    frame[y0:y1, x0:x1, 0] = sub
    frame[y0:y1, x0:x1, 1] = (3 * sub) >> 2
    frame[y0:y1, x0:x1, 2] = (3 * sub) // 5


def face_frame(w: int, h: int, faces, gray: int = 110, cascade_path: str | None = None) -> np.ndarray:
    """Family F: flat gray frame with faces [(x, y, s), ...]."""
    out = np.empty((h, w, 4), dtype=np.uint8)
    out[..., :3] = gray
    out[..., 3] = 255
    for (x, y, s) in faces:
        paste_face(out, int(x), int(y), int(s), cascade_path)
    return out


def seeded_faces(w: int, h: int, seed: int, nmax: int = 3):
    """1..nmax non-overlapping faces at seeded positions, sizes in [32, min(w,h)/2]."""
    r = lcg_stream(seed * 2654435761 & _M32, 64).astype(np.int64) >> 8
    k = 0
    n = 1 + int(r[k] % nmax); k += 1
    smax = max(33, min(w, h) // 2)
    faces = []
    tries = 0
    while len(faces) < n and tries < 16 and k + 3 < len(r):
        s = 32 + int(r[k] % (smax - 32 + 1)); x = int(r[k + 1] % max(1, w - s)); y = int(r[k + 2] % max(1, h - s))
        k += 3; tries += 1
        if all(x + s + 8 <= fx or fx + fs + 8 <= x or y + s + 8 <= fy or fy + fs + 8 <= y for fx, fy, fs in faces):
            faces.append((x, y, s))
    return faces


def mixed_batch(n: int, w: int, h: int, seed0: int = 1234) -> np.ndarray:
    """Benchmark batch: frame i is N / S / F for i % 3 == 0 / 1 / 2, seed = seed0 + i."""
    out = np.empty((n, h, w, 4), dtype=np.uint8)
    for i in range(n):
        seed = seed0 + i
        if i % 3 == 0:
            out[i] = noise_frame(w, h, seed)
        elif i % 3 == 1:
            out[i] = smooth_frame(w, h, seed)
        else:
            out[i] = face_frame(w, h, seeded_faces(w, h, seed))
    return out


def blob_frame(w: int, h: int, cx: int, cy: int, a: int, b: int, rot=(1, 0, 1), color=(200, 60, 40), seed: int = 7,
               bg: str = "noise") -> np.ndarray:
    """Camshift target: a filled ellipse (semi-axes a, b; rotation given as integer (cos_num, sin_num, den), e.g.
    (4, 3, 5)) of a saturated colour with +-7 per-channel seeded jitter, on a noise or flat-gray background."""
    out = noise_frame(w, h, seed) if bg == "noise" else face_frame(w, h, [], gray=110)
    cn, sn, den = rot
    x = np.arange(w, dtype=np.int64)[None, :] - cx
    y = np.arange(h, dtype=np.int64)[:, None] - cy
    u = x * cn + y * sn      # scaled by den
    v = -x * sn + y * cn
    inside = (u * u) * (b * b) + (v * v) * (a * a) <= (a * a) * (b * b) * (den * den)
    jit = (lcg_stream(seed + 977, 3 * w * h) >> np.uint32(28)).astype(np.int64).reshape(h, w, 3) - 8  # -8..7
    for c in range(3):
        ch = out[..., c].astype(np.int64)
        ch = np.where(inside, np.clip(color[c] + jit[..., c], 0, 255), ch)
        out[..., c] = ch.astype(np.uint8)
    return out


def make(gen: dict, w: int, h: int) -> np.ndarray:
    """Build a frame from a generator spec (the specs are stored in tests/golden/*.json next to the expected output)."""
    fam = gen["family"]
    if fam == "noise":
        return noise_frame(w, h, int(gen["seed"]))
    if fam == "smooth":
        return smooth_frame(w, h, int(gen["seed"]))
    if fam == "face":
        return face_frame(w, h, gen["faces"], gray=int(gen.get("gray", 110)))
    if fam == "blob":
        return blob_frame(w, h, int(gen["cx"]), int(gen["cy"]), int(gen["a"]), int(gen["b"]), tuple(gen.get("rot", (1, 0, 1))),
                          tuple(gen.get("color", (200, 60, 40))), int(gen.get("seed", 7)), gen.get("bg", "noise"))
    if fam == "mixed":
        i = int(gen["index"]); seed = int(gen.get("seed0", 1234)) + i
        if i % 3 == 0:
            return noise_frame(w, h, seed)
        if i % 3 == 1:
            return smooth_frame(w, h, seed)
        return face_frame(w, h, seeded_faces(w, h, seed))
    raise ValueError(f"unknown frame family {fam!r}")


def stream_feed_frames(nuniq: int, w: int, h: int, rank: int = 0) -> np.ndarray:
    """The C5 feeds of bench.py (BASELINE.json configs[4]): `nuniq` distinct frames of one camera — a 360-px face drifting
    3 px right / 1 px down per frame over a flat background.  Feed f at time step k shows frame stream_frame_index(k, f, nuniq)."""
    out = np.empty((nuniq, h, w, 4), dtype=np.uint8)
    for k in range(nuniq):
        out[k] = face_frame(w, h, [(700 + 3 * k + 40 * rank, 300 + k, 360)])
    return out


def stream_frame_index(step: int, feed: int, nuniq: int) -> int:
    """feed f runs 7 f frames ahead of feed 0; the cycle repeats every nuniq steps"""
    return (step % nuniq + 7 * feed) % nuniq
