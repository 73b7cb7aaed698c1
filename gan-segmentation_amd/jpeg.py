"""ctypes binding of the on-device baseline JPEG encoder (include/gsa_jpeg.h, csrc/gsa_jpeg.hip) -- the image
half of the dataset writer (reference main.py:100-101 ``cv2.imwrite(img_%06d.jpg)``).

``JpegEncoder(n, H, W)`` owns the device workspace and output buffers for batches of up to ``n`` images;
``encode(img)`` enqueues the kernels on the current stream of ``img`` and returns device tensors
``(scan (n, stride) u8, lengths (n,) i32)``; a file is ``encoder.header + scan[i, :lengths[i]]``.  No CPU fallback."""
import ctypes

import torch

from . import _lib
from ._runtime import current_stream_ptr

_FUNCS = None

DEFAULT_QUALITY = 95     # cv2.imwrite's default JPEG quality (the reference passes none)
DEFAULT_RESTART = 4      # MCUs (16x16 px) per restart interval = one wave of the entropy-coding kernel: 1024 independent
#                          waves per 1024^2 image, +0.25 % bytes; measured on 8 FFHQ-size images: 0.09 ms for restart 1, 2 or 4


def _api():
    global _FUNCS
    if _FUNCS is None:
        lib = _lib.load_library().lib
        c = ctypes
        vp, i32, i64 = c.c_void_p, c.c_int32, c.c_int64
        sig = {
            "gsa_jpeg_header": (i64, [i32, i32, i32, i32, vp, i64]),
            "gsa_jpeg_workspace_bytes": (i64, [i32, i32, i32, i32]),
            "gsa_jpeg_max_scan_bytes": (i64, [i32, i32, i32]),
            "gsa_jpeg_encode": (c.c_int, [vp, i32, i32, i32, vp, i32, i32, vp, i64, vp, i64, vp]),
        }
        _FUNCS = {}
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
            _FUNCS[name] = fn
    return _FUNCS


def header(H, W, quality=DEFAULT_QUALITY, restart=DEFAULT_RESTART):
    """The bytes in front of the entropy-coded data (SOI .. SOS), host side."""
    buf = ctypes.create_string_buffer(1024)
    n = _api()["gsa_jpeg_header"](H, W, quality, restart, ctypes.cast(buf, ctypes.c_void_p), 1024)
    if n < 0 or n > 1024:
        raise _lib.GsaError("gsa_jpeg_header failed (%d)" % n)
    return buf.raw[:n]


class JpegEncoder:
    def __init__(self, n, H, W, device, quality=DEFAULT_QUALITY, restart=DEFAULT_RESTART, out_stride=None):
        api = _api()
        self.n, self.H, self.W, self.quality, self.restart = n, H, W, quality, restart
        self.device = torch.device(device)
        ws = api["gsa_jpeg_workspace_bytes"](n, H, W, restart)
        worst = api["gsa_jpeg_max_scan_bytes"](H, W, restart)
        if ws < 0 or worst < 0:
            raise ValueError("JPEG encoder: images must be multiples of 16 px (got %dx%d), restart in 1..65535; other "
                             "sizes go through the host encoder (DatasetWriter(gpu_jpeg=False) / JPEG_ON_GPU: false)" % (H, W))
        self.header = header(H, W, quality, restart)
        # default stride: the size of the raw pixels (a q95 scan is ~1/7 of it; noise at q100 can exceed it -> the
        # call reports the size needed as a negative length and encode() retries with the worst-case stride)
        self.out_stride = int(min(worst, out_stride if out_stride is not None else H * W * 3))
        self.worst = int(worst)
        self._ws = torch.empty(ws, dtype=torch.uint8, device=self.device)
        self._alloc_out()

    def _alloc_out(self):
        self.out = torch.empty((self.n, self.out_stride), dtype=torch.uint8, device=self.device)
        self.lengths = torch.empty(self.n, dtype=torch.int32, device=self.device)

    def encode(self, img):
        """img: (k, H, W, 3) uint8 CUDA tensor, k <= n.  -> (scan (k, stride) u8, lengths (k,) i32), stream-ordered."""
        if not img.is_cuda or img.dtype != torch.uint8 or not img.is_contiguous():
            raise ValueError("encode takes a contiguous uint8 CUDA tensor")
        k = img.shape[0]
        if k > self.n or tuple(img.shape[1:]) != (self.H, self.W, 3):
            raise ValueError("image batch %s does not fit the encoder (%d, %d, %d, 3)" % (tuple(img.shape), self.n, self.H, self.W))
        with torch.cuda.device(img.device):     # the C ABI is stateless: kernels go to the calling thread's current device
            rc = _api()["gsa_jpeg_encode"](current_stream_ptr(img.device), k, self.H, self.W, img.data_ptr(), self.quality,
                                           self.restart, self._ws.data_ptr(), self._ws.numel(), self.out.data_ptr(),
                                           self.out_stride, self.lengths.data_ptr())
        if rc != 0:
            raise _lib.GsaError("gsa_jpeg_encode failed (%d)" % rc)
        return self.out[:k], self.lengths[:k]

    def grow(self):
        """Switch to the worst-case stride (after a negative length)."""
        self.out_stride = self.worst
        self._alloc_out()

    def files(self, img):
        """Convenience (tests, small jobs): encode and return the complete files as ``bytes`` (synchronises)."""
        scan, lengths = self.encode(img)
        ln = lengths.cpu().numpy()
        if (ln < 0).any():
            self.grow()
            scan, lengths = self.encode(img)
            ln = lengths.cpu().numpy()
        host = scan.cpu().numpy()
        return [self.header + host[i, :ln[i]].tobytes() for i in range(len(ln))]
