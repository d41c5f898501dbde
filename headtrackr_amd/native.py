"""ctypes binding of libheadtrackr_hip.so (include/headtrackr_hip.h) — the only compute path of this package.

There is deliberately no CPU fallback: if the shared library is missing or no gfx950 device is visible, importing
`lib()` / creating a context raises.  (The CPU checker lives in oracle/ and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEADTRACKR_HIP_LIB: load another BUILD of the same library (an instrumented variant from tools/build_alt.py) — a loader path like
# LD_LIBRARY_PATH, not a behaviour switch: the library itself reads no environment variable
LIB_PATH = os.environ.get("HEADTRACKR_HIP_LIB") or os.path.join(_HERE, "libheadtrackr_hip.so")

HT_OK = 0
HT_ERR_CAPACITY = -4
HT_INPUT_RGBA = 0
HT_INPUT_GRAY_IN_R = 1
HT_SCAN_NO_SPLIT = 2
HT_SCAN_SIMPLE = 4
HT_SCAN_GENERIC = 8
HT_SCAN_STATS = 16
HT_DETECT_WHITEBALANCE = 32
HT_MAX_LEVELS = 96


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("interval", C.c_int32),
        ("hit_capacity", C.c_uint32),
        ("stream", C.c_void_p),
        ("queue_capacity", C.c_uint32),
        ("flags", C.c_uint32),
        ("options", C.c_char_p),  # "key=value,...": per-context schedule selectors (include/headtrackr_hip.h); None = defaults
    ]


HIT_DTYPE = np.dtype(
    [("frame", "<u4"), ("x", "<u2"), ("y", "<u2"), ("scale", "u1"), ("q", "u1"), ("reserved0", "<u2"), ("reserved1", "<u4"), ("sum", "<f8")]
)
RECT_DTYPE = np.dtype(
    [("x", "<f8"), ("y", "<f8"), ("width", "<f8"), ("height", "<f8"), ("confidence", "<f8"), ("neighbors", "<i4"), ("reserved", "<i4")]
)
CS_RECT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4")])
CS_TRACKOBJ_DTYPE = np.dtype(
    [("x", "<f8"), ("y", "<f8"), ("width", "<f8"), ("height", "<f8"), ("angle", "<f8"),
     ("sw_x", "<i4"), ("sw_y", "<i4"), ("sw_width", "<i4"), ("sw_height", "<i4")]
)
KERNEL_TIME_DTYPE = np.dtype([("name", "S32"), ("ms", "<f8"), ("launches", "<u4"), ("reserved", "<u4")])
assert HIT_DTYPE.itemsize == 24 and RECT_DTYPE.itemsize == 48 and CS_TRACKOBJ_DTYPE.itemsize == 56 and KERNEL_TIME_DTYPE.itemsize == 48


class PlaneInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32), ("present", C.c_int32), ("offset", C.c_uint64)]


# every symbol include/headtrackr_hip.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "ht_create", "ht_destroy", "ht_last_error", "ht_abi_version", "ht_set_geometry", "ht_num_levels", "ht_plane",
    "ht_windows_per_frame", "ht_pyramid_bytes_per_frame", "ht_upload_frames", "ht_upload_frames_async", "ht_swap_frames", "ht_bind_frames_device", "ht_frames_bound", "ht_frames_enqueued", "ht_host_alloc", "ht_host_free", "ht_device_alloc", "ht_device_free", "ht_device_upload", "ht_detect_enqueue",
    "ht_detect_collect", "ht_detect_batch", "ht_pyramid_readback", "ht_stage_counts", "ht_grayscale_batch",
    "ht_whitebalance_batch", "ht_detect_whitebalance", "ht_hits_to_rects", "ht_group_rects", "ht_best_faces", "ht_detect_collect_best", "ht_detect_collect_best_requeue", "ht_camshift_reserve", "ht_camshift_init_batch",
    "ht_camshift_track_batch", "ht_camshift_track_collect", "ht_camshift_track_sequence", "ht_camshift_sequence_collect", "ht_camshift_stats", "ht_camshift_debug_hist", "ht_allgather_records", "ht_allgather_best_faces", "ht_device_count", "ht_profile", "ht_kernel_times", "ht_stream", "ht_graph_launches", "ht_synchronize",
]

_lib = None


def lib():
    """Loads libheadtrackr_hip.so; raises if it has not been built (python -m headtrackr_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m headtrackr_amd.build` (no CPU fallback exists)")
    L = C.CDLL(LIB_PATH)
    vp, u8p, i32, u32, sz = C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t
    L.ht_create.restype = i32
    L.ht_create.argtypes = [C.POINTER(Config), vp, sz, C.POINTER(vp)]
    L.ht_destroy.restype = None
    L.ht_destroy.argtypes = [vp]
    L.ht_last_error.restype = C.c_char_p
    L.ht_last_error.argtypes = [vp]
    L.ht_abi_version.restype = i32
    L.ht_set_geometry.restype = i32
    L.ht_set_geometry.argtypes = [vp, i32, i32, i32, vp, i32]
    L.ht_num_levels.restype = i32
    L.ht_num_levels.argtypes = [vp]
    L.ht_plane.restype = i32
    L.ht_plane.argtypes = [vp, i32, i32, C.POINTER(PlaneInfo)]
    L.ht_windows_per_frame.restype = C.c_uint64
    L.ht_windows_per_frame.argtypes = [vp]
    L.ht_pyramid_bytes_per_frame.restype = C.c_uint64
    L.ht_pyramid_bytes_per_frame.argtypes = [vp]
    L.ht_upload_frames.restype = i32
    L.ht_upload_frames.argtypes = [vp, u8p, i32, sz]
    L.ht_upload_frames_async.restype = i32
    L.ht_upload_frames_async.argtypes = [vp, u8p, i32, sz]
    L.ht_swap_frames.restype = i32
    L.ht_swap_frames.argtypes = [vp]
    L.ht_bind_frames_device.restype = i32
    L.ht_bind_frames_device.argtypes = [vp, vp, i32, sz]
    L.ht_frames_bound.restype = i32
    L.ht_frames_bound.argtypes = [vp]
    L.ht_frames_enqueued.restype = i32
    L.ht_frames_enqueued.argtypes = [vp]
    L.ht_host_alloc.restype = i32
    L.ht_host_alloc.argtypes = [sz, C.POINTER(vp)]
    L.ht_host_free.restype = None
    L.ht_host_free.argtypes = [vp]
    L.ht_device_alloc.restype = i32
    L.ht_device_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.ht_device_free.restype = i32
    L.ht_device_free.argtypes = [vp, vp]
    L.ht_device_upload.restype = i32
    L.ht_device_upload.argtypes = [vp, vp, vp, sz]
    L.ht_detect_enqueue.restype = i32
    L.ht_detect_enqueue.argtypes = [vp, u32]
    L.ht_detect_collect.restype = i32
    L.ht_detect_collect.argtypes = [vp, vp, u32, vp, C.POINTER(u32)]
    L.ht_detect_batch.restype = i32
    L.ht_detect_batch.argtypes = [vp, u8p, i32, i32, i32, sz, u32, vp, u32, vp, C.POINTER(u32)]
    L.ht_pyramid_readback.restype = i32
    L.ht_pyramid_readback.argtypes = [vp, i32, i32, i32, u8p, sz]
    L.ht_stage_counts.restype = i32
    L.ht_stage_counts.argtypes = [vp, vp, i32]
    L.ht_grayscale_batch.restype = i32
    L.ht_grayscale_batch.argtypes = [vp, u8p, i32, i32, i32, sz]
    L.ht_whitebalance_batch.restype = i32
    L.ht_whitebalance_batch.argtypes = [vp, vp, i32]
    L.ht_hits_to_rects.restype = i32
    L.ht_hits_to_rects.argtypes = [vp, vp, u32, vp]
    L.ht_group_rects.restype = i32
    L.ht_group_rects.argtypes = [vp, u32, i32, vp, C.POINTER(u32)]
    L.ht_best_faces.restype = i32
    L.ht_best_faces.argtypes = [vp, vp, vp, i32, i32, vp]
    L.ht_detect_collect_best.restype = i32
    L.ht_detect_collect_best.argtypes = [vp, i32, vp, C.POINTER(u32)]
    L.ht_detect_collect_best_requeue.restype = i32
    L.ht_detect_collect_best_requeue.argtypes = [vp, i32, vp, vp, C.c_uint32]
    L.ht_camshift_reserve.restype = i32
    L.ht_camshift_reserve.argtypes = [vp, i32]
    L.ht_camshift_init_batch.restype = i32
    L.ht_camshift_init_batch.argtypes = [vp, i32, i32, vp]
    L.ht_camshift_track_batch.restype = i32
    L.ht_camshift_track_batch.argtypes = [vp, i32, i32, i32, vp]
    L.ht_detect_whitebalance.restype = i32
    L.ht_detect_whitebalance.argtypes = [vp, vp, i32]
    L.ht_camshift_track_sequence.restype = i32
    L.ht_camshift_track_sequence.argtypes = [vp, i32, i32, i32, vp, i32, sz, vp, i32]
    L.ht_camshift_sequence_collect.restype = i32
    L.ht_camshift_sequence_collect.argtypes = [vp, i32, i32, i32, vp]
    L.ht_camshift_stats.restype = i32
    L.ht_camshift_stats.argtypes = [vp, i32, i32, vp, vp, i32]
    L.ht_camshift_debug_hist.restype = i32
    L.ht_camshift_debug_hist.argtypes = [vp, i32, vp, vp]
    L.ht_allgather_records.restype = i32
    L.ht_allgather_records.argtypes = [vp, i32, vp, sz]
    L.ht_allgather_best_faces.restype = i32
    L.ht_allgather_best_faces.argtypes = [vp, i32, vp, i32, vp]
    L.ht_device_count.restype = i32
    L.ht_device_count.argtypes = []
    L.ht_profile.restype = i32
    L.ht_profile.argtypes = [vp, i32]
    L.ht_kernel_times.restype = i32
    L.ht_kernel_times.argtypes = [vp, vp, C.POINTER(i32), i32]
    L.ht_stream.restype = vp
    L.ht_stream.argtypes = [vp]
    L.ht_camshift_track_collect.restype = i32
    L.ht_camshift_track_collect.argtypes = [vp, i32, vp]
    L.ht_graph_launches.restype = C.c_uint64
    L.ht_graph_launches.argtypes = [vp]
    L.ht_synchronize.restype = i32
    L.ht_synchronize.argtypes = [vp]
    _lib = L
    return L


class HtError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"headtrackr_hip status {status}: {msg}")
        self.status = status
