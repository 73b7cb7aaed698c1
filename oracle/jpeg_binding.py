"""TEST INFRASTRUCTURE ONLY -- numpy front end of the scalar JPEG oracle (oracle/c/jpeg_oracle.c).
Only tests/ may import this; the product path is gan_segmentation_amd.jpeg (HIP, no CPU fallback)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY = os.path.join(_HERE, "c", "libgsa_jpeg_oracle.so")
_lib = None


def _api():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "c", "jpeg_oracle.c")
        if not os.path.exists(LIBRARY) or os.path.getmtime(LIBRARY) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "c/libgsa_jpeg_oracle.so"], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(LIBRARY)
        i32, i64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
        _lib.gsao_jpeg_encode.restype, _lib.gsao_jpeg_encode.argtypes = i64, [i32, i32, vp, i32, i32, vp, i64]
        _lib.gsao_jpeg_header.restype, _lib.gsao_jpeg_header.argtypes = i64, [i32, i32, i32, i32, vp, i64]
    return _lib


def encode(img, quality=95, restart=0):
    """img (H,W,3) u8 -> the complete JPEG file as bytes."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W, _ = img.shape
    cap = H * W * 6 + 4096
    out = np.empty(cap, np.uint8)
    n = _api().gsao_jpeg_encode(H, W, img.ctypes.data, quality, restart, out.ctypes.data, cap)
    if n < 0:
        raise ValueError("jpeg oracle: bad arguments")
    if n > cap:
        out = np.empty(n, np.uint8)
        n = _api().gsao_jpeg_encode(H, W, img.ctypes.data, quality, restart, out.ctypes.data, n)
    return out[:n].tobytes()


def header(H, W, quality=95, restart=0):
    out = np.empty(1024, np.uint8)
    n = _api().gsao_jpeg_header(H, W, quality, restart, out.ctypes.data, 1024)
    return out[:n].tobytes()
