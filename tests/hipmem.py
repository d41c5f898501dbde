"""Device buffers for tests that hand device-resident frames to the C ABI (ht_bind_frames_device, ht_camshift_track_sequence):
plain hipMalloc / hipMemcpy through ctypes on the HIP runtime the library itself is linked against — no torch in the tests."""
import ctypes as C

import numpy as np

_hip = None


def _rt():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64 not found")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipFree.argtypes = [C.c_void_p]
    return _hip


class DeviceArray:
    def __init__(self, host: np.ndarray):
        host = np.ascontiguousarray(host)
        p = C.c_void_p()
        if _rt().hipMalloc(C.byref(p), host.nbytes) != 0:
            raise MemoryError("hipMalloc failed")
        self.ptr, self.nbytes = p.value, host.nbytes
        if _rt().hipMemcpy(self.ptr, host.ctypes.data, host.nbytes, 1) != 0:  # hipMemcpyHostToDevice
            raise RuntimeError("hipMemcpy failed")

    @classmethod
    def tiled(cls, base: np.ndarray, n: int) -> "DeviceArray":
        """n frames on the device, frame i = base[i % len(base)], without building the n-frame array on the host
        (1024 x 1280x720 RGBA = 3.8 GB)."""
        base = np.ascontiguousarray(base)
        fb = base[0].nbytes
        self = cls.__new__(cls)
        p = C.c_void_p()
        if _rt().hipMalloc(C.byref(p), fb * n) != 0:
            raise MemoryError("hipMalloc failed")
        self.ptr, self.nbytes = p.value, fb * n
        for i in range(n):
            if _rt().hipMemcpy(self.ptr + i * fb, base[i % len(base)].ctypes.data, fb, 1) != 0:
                raise RuntimeError("hipMemcpy failed")
        return self

    def free(self):
        if self.ptr:
            _rt().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
