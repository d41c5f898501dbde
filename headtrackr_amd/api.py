"""Host-side Python mirror of the reference's per-frame interface, on top of the C ABI (include/headtrackr_hip.h).

The primary host language of this project is JavaScript (headtrackr_amd/js/headtrackr.js over the N-API addon); this
module exposes the same operations to Python so that the parity tests (pytest) and bench.py can drive the same C ABI.
Names follow the reference: ccv.grayscale / ccv.detect_objects (src/ccv.js:22,109), camshift.Tracker
(src/camshift.js:148), getWhitebalance (src/whitebalance.js:5).  Every method calls the HIP library; none computes on
the CPU except ccv's own O(n^2) grouping of a few dozen rectangles, which the C ABI keeps on the host by design.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import native
from .cascade import Cascade, load_cascade
from .native import (HIT_DTYPE, HT_INPUT_GRAY_IN_R, HT_INPUT_RGBA, HT_SCAN_NO_SPLIT, HT_SCAN_SIMPLE, RECT_DTYPE,  # noqa: F401
                     HtError)


class Context:
    """One ht_ctx: one GPU, one cascade, one stream."""

    def __init__(self, cascade: Cascade | None = None, device: int = 0, interval: int = 5, stream: int | None = None,
                 hit_capacity: int = 0, queue_capacity: int = 0, options: str | dict | None = None):
        """options: ht_config.options — "key=value,..." or a dict (schedule selectors for tests / A-B runs; results never change)"""
        self._lib = native.lib()
        self.cascade = cascade or load_cascade()
        if isinstance(options, dict):
            options = ",".join(f"{k}={int(v)}" for k, v in options.items())
        cfg = native.Config(C.sizeof(native.Config), device, interval, hit_capacity, stream, queue_capacity, 0, options.encode() if options else None)
        h = C.c_void_p()
        blob = self.cascade.blob
        st = self._lib.ht_create(C.byref(cfg), blob, len(blob), C.byref(h))
        if st != 0:
            raise HtError(st, self._lib.ht_last_error(None).decode())
        self._h = h
        self.interval = interval
        self.width = self.height = 0
        self._collected_n = 0
        self.nframes = 0   # frames currently bound (upload / bind_device / swap_frames)

    # -- plumbing -------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.ht_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int, allow=()):
        if st != 0 and st not in allow:
            raise HtError(st, self._lib.ht_last_error(self._h).decode())
        return st

    # -- geometry / frames ------------------------------------------------------------------------------------
    def set_geometry(self, width: int, height: int, max_batch: int, level_dims=None):
        if level_dims is not None:
            ld = np.ascontiguousarray(level_dims, dtype=np.int32).reshape(-1)
            self._check(self._lib.ht_set_geometry(self._h, width, height, max_batch, ld.ctypes.data, ld.size // 2))
        else:
            self._check(self._lib.ht_set_geometry(self._h, width, height, max_batch, None, 0))
        self.width, self.height = width, height

    @property
    def num_levels(self) -> int:
        return self._lib.ht_num_levels(self._h)

    @property
    def windows_per_frame(self) -> int:
        return self._lib.ht_windows_per_frame(self._h)

    @property
    def pyramid_bytes_per_frame(self) -> int:
        return self._lib.ht_pyramid_bytes_per_frame(self._h)

    def plane(self, level: int, slot: int = 0):
        p = native.PlaneInfo()
        self._check(self._lib.ht_plane(self._h, level, slot, C.byref(p)))
        return p

    def upload(self, frames: np.ndarray):
        """frames: uint8 [n, H, W, 4] host array."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, h, w, c = frames.shape
        assert c == 4 and (w, h) == (self.width, self.height), "call set_geometry(w, h, max_batch) first"
        self._check(self._lib.ht_upload_frames(self._h, frames.ctypes.data, n, w * h * 4))
        self.nframes = n

    def upload_ptr(self, host_ptr: int, n: int, frame_stride: int | None = None):
        """Like upload(), from a raw host pointer (e.g. pinned memory: the copy then runs at PCIe speed)."""
        self._check(self._lib.ht_upload_frames(self._h, host_ptr, n, frame_stride or self.width * self.height * 4))
        self.nframes = n

    def upload_async_ptr(self, host_ptr: int, n: int, frame_stride: int | None = None):
        """Starts copying the NEXT n frames from pinned host memory into the back buffer on the copy stream; the kernels of
        the current frames keep running.  swap_frames() makes them current."""
        self._check(self._lib.ht_upload_frames_async(self._h, host_ptr, n, frame_stride or self.width * self.height * 4))
        self._back_n = n

    def swap_frames(self):
        self._check(self._lib.ht_swap_frames(self._h))
        self.nframes = self._back_n

    def bind_device(self, dev_ptr: int, n: int, frame_stride: int | None = None):
        """Use n RGBA frames already resident in device memory (e.g. a torch cuda tensor's data_ptr())."""
        self._check(self._lib.ht_bind_frames_device(self._h, dev_ptr, n, frame_stride or self.width * self.height * 4))
        self.nframes = n

    # -- detect -----------------------------------------------------------------------------------------------
    def detect_enqueue(self, flags: int = HT_INPUT_RGBA):
        self._check(self._lib.ht_detect_enqueue(self._h, flags))

    # The library sizes what it writes (counts[], best[]) by the batch that was ENQUEUED, not by what is bound at collect time
    # (a streaming host binds or swaps in the next frames in between): the buffers below follow ht_frames_enqueued.
    def detect_collect(self, cap: int = 1 << 16):
        n = self._collected_n = int(self._lib.ht_frames_enqueued(self._h))  # the library's own count of the batch in flight
        buf = getattr(self, "_hitbuf", None)
        if buf is None or len(buf) < cap:
            buf = self._hitbuf = np.empty(cap, dtype=HIT_DTYPE)  # reused across calls
        counts = np.empty(max(1, n), dtype=np.uint32)
        total = C.c_uint32(0)
        self._check(self._lib.ht_detect_collect(self._h, buf.ctypes.data, cap, counts.ctypes.data, C.byref(total)))
        return buf[: total.value].copy(), counts[:n]

    def detect_collect_best(self, min_neighbors: int = 1, out: np.ndarray | None = None):
        """ht_detect_collect + ht_best_faces in one C call: (best rect per frame of the enqueued batch, raw hit count)."""
        n = self._collected_n = int(self._lib.ht_frames_enqueued(self._h))  # the library's own count of the batch in flight
        if out is None or len(out) < n:
            out = np.zeros(max(1, n), dtype=RECT_DTYPE)
        total = C.c_uint32(0)
        self._check(self._lib.ht_detect_collect_best(self._h, min_neighbors, out.ctypes.data, C.byref(total)))
        return out[:n], total.value

    def detect_collect_best_requeue(self, min_neighbors: int = 1, out: np.ndarray | None = None, next_flags: int = HT_INPUT_RGBA):
        """detect_collect_best, and the next batch of the bound frames is enqueued as soon as this batch's raw hits are on the host
        (before they are sorted and grouped)."""
        n = self._collected_n = int(self._lib.ht_frames_enqueued(self._h))  # the library's own count of the batch in flight
        if out is None or len(out) < n:
            out = np.zeros(max(1, n), dtype=RECT_DTYPE)
        total = C.c_uint32(0)
        self._check(self._lib.ht_detect_collect_best_requeue(self._h, min_neighbors, out.ctypes.data, C.byref(total), next_flags))
        return out[:n], total.value

    def detect_raw(self, frames: np.ndarray, flags: int = HT_INPUT_RGBA, cap: int = 1 << 16):
        """ccv.grayscale + ccv.detect_objects(..., min_neighbors = 0) for a batch: (hits, per-frame counts)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim == 3:
            frames = frames[None]
        n, h, w, _ = frames.shape
        if (w, h) != (self.width, self.height) or n > getattr(self, "_max_batch", 0):
            self.set_geometry(w, h, n)
            self._max_batch = n
        self.upload(frames)
        self.detect_enqueue(flags)
        return self.detect_collect(cap)

    def hits_to_rects(self, hits: np.ndarray) -> np.ndarray:
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        out = np.zeros(len(hits), dtype=RECT_DTYPE)
        if len(hits):
            self._check(self._lib.ht_hits_to_rects(self._h, hits.ctypes.data, len(hits), out.ctypes.data))
        return out

    def group_rects(self, rects: np.ndarray, min_neighbors: int = 1) -> np.ndarray:
        rects = np.ascontiguousarray(rects, dtype=RECT_DTYPE)
        out = np.zeros(max(1, len(rects)), dtype=RECT_DTYPE)
        n = C.c_uint32(0)
        self._check(self._lib.ht_group_rects(rects.ctypes.data, len(rects), min_neighbors, out.ctypes.data, C.byref(n)))
        return out[: n.value].copy()

    def best_faces(self, hits: np.ndarray, counts: np.ndarray, min_neighbors: int = 1) -> np.ndarray:
        """facetrackr's choice per frame (facetrackr.js:147-175): the grouped rect with the highest confidence."""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        out = np.zeros(len(counts), dtype=RECT_DTYPE)
        self._check(self._lib.ht_best_faces(self._h, hits.ctypes.data if len(hits) else None, counts.ctypes.data, len(counts), min_neighbors,
                                            out.ctypes.data))
        return out

    def detect_objects(self, frames: np.ndarray, min_neighbors: int = 1, flags: int = HT_INPUT_RGBA):
        """Per frame: the list ccv.detect_objects(canvas, cascade, interval, min_neighbors) returns (ccv.js:109-333)."""
        hits, counts = self.detect_raw(frames, flags)
        out, k = [], 0
        for c in counts:
            r = self.hits_to_rects(hits[k : k + int(c)])
            out.append(self.group_rects(r, min_neighbors) if min_neighbors > 0 else r)
            k += int(c)
        return out

    def pyramid_readback(self, frame: int, level: int, slot: int = 0) -> np.ndarray:
        p = self.plane(level, slot)
        out = np.zeros((p.height, p.width), dtype=np.uint8)
        self._check(self._lib.ht_pyramid_readback(self._h, frame, level, slot, out.ctypes.data, out.size))
        return out

    def stage_counts(self) -> np.ndarray:
        n = self.cascade.count + 1
        out = np.zeros(n, dtype=np.uint64)
        self._check(self._lib.ht_stage_counts(self._h, out.ctypes.data, n))
        return out

    def grayscale(self, frames: np.ndarray) -> np.ndarray:
        """ccv.grayscale for a batch of host RGBA frames; returns the gray RGBA copy."""
        out = np.ascontiguousarray(frames, dtype=np.uint8).copy()
        if out.ndim == 3:
            out = out[None]
        n, h, w, _ = out.shape
        self._check(self._lib.ht_grayscale_batch(self._h, out.ctypes.data, n, w, h, w * h * 4))
        return out

    def detect_whitebalance(self, n: int | None = None) -> np.ndarray:
        """getWhitebalance of the frames of the batch COLLECTED last, if it was enqueued with HT_DETECT_WHITEBALANCE (fused into
        its gray pass; the sums are snapshotted by the collect call, so a batch re-enqueued meanwhile does not disturb them)."""
        n = self._collected_n if n is None else n
        out = np.zeros(max(n, 1), dtype=np.float64)[:n]
        self._check(self._lib.ht_detect_whitebalance(self._h, out.ctypes.data, n))
        return out

    def whitebalance(self) -> np.ndarray:
        """headtrackr.getWhitebalance for every bound frame."""
        out = np.zeros(self.nframes, dtype=np.float64)
        self._check(self._lib.ht_whitebalance_batch(self._h, out.ctypes.data, self.nframes))
        return out

    # -- camshift -----------------------------------------------------------------------------------------------
    def camshift_reserve(self, nstreams: int):
        self._check(self._lib.ht_camshift_reserve(self._h, nstreams))

    def camshift_init(self, rects, first: int = 0):
        r = np.zeros(len(rects), dtype=native.CS_RECT_DTYPE)
        for i, (x, y, w, h) in enumerate(rects):
            r[i] = (x, y, w, h)
        self._check(self._lib.ht_camshift_init_batch(self._h, first, len(rects), r.ctypes.data))

    def camshift_track(self, n: int, calc_angles: bool = True, first: int = 0, fetch: bool = True):
        out = np.zeros(n, dtype=native.CS_TRACKOBJ_DTYPE)
        self._check(self._lib.ht_camshift_track_batch(self._h, first, n, int(calc_angles), out.ctypes.data if fetch else None))
        return out

    def camshift_track_collect(self, n: int):
        """Track objects of the OLDEST outstanding camshift_track(n, fetch=False): waits for that call only.  Up to 4 enqueue-only calls
        may be outstanding (results land in a ring of pinned slots), so a streaming host enqueues step i + 1 before it waits for step i,
        and a host with several feeds (contexts) enqueues every feed's track() first and collects afterwards."""
        out = np.zeros(n, dtype=native.CS_TRACKOBJ_DTYPE)
        self._check(self._lib.ht_camshift_track_collect(self._h, n, out.ctypes.data))
        return out

    def camshift_track_sequence(self, dev_ptrs, n: int, calc_angles: bool = True, first: int = 0, frame_stride: int | None = None,
                                fetch: str = "last", keep_all: bool = False):
        """len(dev_ptrs) successive track() calls in one host call; call k reads the n device-resident frames at dev_ptrs[k].
        fetch: "last" -> [n] track objects of the last call, "all" -> [calls, n], "none" -> enqueue only (camshift_sequence_collect
        fetches later: the last call's objects, or every call's if keep_all)."""
        k = len(dev_ptrs)
        ptrs = (C.c_void_p * k)(*[int(p) for p in dev_ptrs])
        stride = frame_stride or self.width * self.height * 4
        if fetch == "none":
            self._check(self._lib.ht_camshift_track_sequence(self._h, first, n, int(calc_angles), ptrs, k, stride, None, int(keep_all)))
            return None
        out = np.zeros((k, n) if fetch == "all" else (n,), dtype=native.CS_TRACKOBJ_DTYPE)
        self._check(self._lib.ht_camshift_track_sequence(self._h, first, n, int(calc_angles), ptrs, k, stride, out.ctypes.data, int(fetch == "all")))
        return out

    def camshift_sequence_collect(self, n: int, ncalls: int, fetch: str = "last"):
        """Results of the last camshift_track_sequence(..., fetch="none"): waits for it ("last" -> [n], "all" -> [ncalls, n])."""
        out = np.zeros((ncalls, n) if fetch == "all" else (n,), dtype=native.CS_TRACKOBJ_DTYPE)
        self._check(self._lib.ht_camshift_sequence_collect(self._h, n, ncalls, int(fetch == "all"), out.ctypes.data))
        return out

    def camshift_stats(self, n: int, first: int = 0, reset: bool = True):
        """(window pixels read by the moment passes, track() calls) per stream since the last reset (SURVEY.md 8d B_track)."""
        px = np.zeros(n, dtype=np.uint64)
        calls = np.zeros(n, dtype=np.uint64)
        self._check(self._lib.ht_camshift_stats(self._h, first, n, px.ctypes.data, calls.ctypes.data, int(reset)))
        return px, calls

    def camshift_debug_hist(self, stream: int, current: bool = True):
        """(model histogram, full-frame histogram of the last track() call) of one stream, 4096 bins each (test hook)."""
        model = np.zeros(4096, dtype=np.uint32)
        cur = np.zeros(4096, dtype=np.uint32)
        self._check(self._lib.ht_camshift_debug_hist(self._h, stream, model.ctypes.data, cur.ctypes.data if current else None))
        return model, cur

    # -- measurement --------------------------------------------------------------------------------------------
    def profile(self, on: bool = True):
        self._check(self._lib.ht_profile(self._h, int(on)))

    def kernel_times(self, reset: bool = True) -> dict:
        # the entry count first (no buffer, nothing reset), then a buffer that holds them all: with a fixed 32 entries the pseudo-timers
        # the library appends behind the real ones (cs_fused_launches_*) fell off the end once many timers existed
        n = C.c_int32(0)
        self._check(self._lib.ht_kernel_times(self._h, None, C.byref(n), 0))
        cap = max(int(n.value), 1)
        buf = np.zeros(cap, dtype=native.KERNEL_TIME_DTYPE)
        n = C.c_int32(cap)
        self._check(self._lib.ht_kernel_times(self._h, buf.ctypes.data, C.byref(n), int(reset)))
        if n.value > cap:
            raise RuntimeError(f"ht_kernel_times: {n.value} entries for a buffer of {cap}")
        return {b["name"].decode(): dict(ms=float(b["ms"]), launches=int(b["launches"])) for b in buf[: n.value]}

    @property
    def graph_launches(self) -> int:
        """detect_enqueue calls that were served by replaying a captured hipGraph"""
        return int(self._lib.ht_graph_launches(self._h))

    def synchronize(self):
        self._check(self._lib.ht_synchronize(self._h))

    @property
    def stream(self) -> int:
        return int(self._lib.ht_stream(self._h) or 0)


def device_count() -> int:
    return int(native.lib().ht_device_count())


def allgather_best_faces(ctxs, best_per_rank) -> np.ndarray:
    """Single-process multi-GPU exchange (ht_allgather_best_faces): ctxs[i] lives on GPU i and produced best_per_rank[i]
    (ht_best_faces output, equal lengths).  Returns the gathered [nranks, frames_per_rank] table after every rank's copy was
    checked to be identical."""
    n = len(ctxs)
    per = len(best_per_rank[0])
    arrs = [np.ascontiguousarray(b, dtype=RECT_DTYPE) for b in best_per_rank]
    assert all(len(a) == per for a in arrs)
    cp = (C.c_void_p * n)(*[c._h.value for c in ctxs])
    bp = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    out = np.zeros((n, per), dtype=RECT_DTYPE)
    st = native.lib().ht_allgather_best_faces(cp, n, bp, per, out.ctypes.data)
    if st != 0:
        raise HtError(st, native.lib().ht_last_error(ctxs[0]._h).decode())
    return out
