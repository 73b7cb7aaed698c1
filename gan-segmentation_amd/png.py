"""ctypes binding of the on-device PNG compressor (include/gsa_png.h, csrc/gsa_png.hip) -- the mask half of the dataset
writer (reference main.py:102-103 ``cv2.imwrite(mask_%06d.png)``).

``PngEncoder(n, H, W)`` owns the device workspace and output buffers; ``encode(mask)`` enqueues the kernels on the
current stream and returns device tensors ``(stream (n, stride) u8, lengths (n,) i32)`` holding each mask's zlib
stream; ``png_file(H, W, stream_bytes)`` adds the PNG chunk framing on the host.  No CPU fallback."""
import ctypes
import struct
import zlib

import torch

from . import _lib
from ._runtime import current_stream_ptr

_FUNCS = None
_SIGNATURE = b"\x89PNG\r\n\x1a\n"


def _api():
    global _FUNCS
    if _FUNCS is None:
        lib = _lib.load_library().lib
        c = ctypes
        vp, i32, i64 = c.c_void_p, c.c_int32, c.c_int64
        sig = {
            "gsa_png_workspace_bytes": (i64, [i32, i32, i32]),
            "gsa_png_max_stream_bytes": (i64, [i32, i32]),
            "gsa_png_encode": (c.c_int, [vp, i32, i32, i32, vp, vp, i64, vp, i64, vp]),
        }
        _FUNCS = {}
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
            _FUNCS[name] = fn
    return _FUNCS


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + bytes(data) + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)


def png_file(H, W, idat):
    """8-bit greyscale PNG around one zlib stream (the IDAT payload the GPU produced)."""
    return _SIGNATURE + _chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", idat) + _chunk(b"IEND", b"")


class PngEncoder:
    def __init__(self, n, H, W, device):
        api = _api()
        self.n, self.H, self.W = n, H, W
        self.device = torch.device(device)
        ws = api["gsa_png_workspace_bytes"](n, H, W)
        worst = api["gsa_png_max_stream_bytes"](H, W)
        if ws < 0 or worst < 0:
            raise ValueError("PNG encoder: the width must be a multiple of 16 px (got %dx%d); other sizes go through the "
                             "host encoder (DatasetWriter(gpu_png=False))" % (H, W))
        self.out_stride = int(worst)             # the worst case (every byte a 9-bit literal) is only 1.13 B/px
        self._ws = torch.empty(ws, dtype=torch.uint8, device=self.device)
        self.out = torch.empty((n, self.out_stride), dtype=torch.uint8, device=self.device)
        self.lengths = torch.empty(n, dtype=torch.int32, device=self.device)

    def encode(self, mask):
        """mask: (k, H, W) uint8 CUDA tensor, k <= n.  -> (zlib streams (k, stride) u8, lengths (k,) i32), stream-ordered."""
        if not mask.is_cuda or mask.dtype != torch.uint8 or not mask.is_contiguous():
            raise ValueError("encode takes a contiguous uint8 CUDA tensor")
        k = mask.shape[0]
        if k > self.n or tuple(mask.shape[1:]) != (self.H, self.W):
            raise ValueError("mask batch %s does not fit the encoder (%d, %d, %d)" % (tuple(mask.shape), self.n, self.H, self.W))
        with torch.cuda.device(mask.device):     # the C ABI is stateless: kernels go to the calling thread's current device
            rc = _api()["gsa_png_encode"](current_stream_ptr(mask.device), k, self.H, self.W, mask.data_ptr(), self._ws.data_ptr(),
                                          self._ws.numel(), self.out.data_ptr(), self.out_stride, self.lengths.data_ptr())
        if rc != 0:
            raise _lib.GsaError("gsa_png_encode failed (%d)" % rc)
        return self.out[:k], self.lengths[:k]

    def files(self, mask):
        """Convenience (tests, small jobs): encode and return the complete PNG files as ``bytes`` (synchronises)."""
        streams, lengths = self.encode(mask)
        ln = lengths.cpu().numpy()
        host = streams.cpu().numpy()
        return [png_file(self.H, self.W, host[i, :ln[i]].tobytes()) for i in range(len(ln))]
