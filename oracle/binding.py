"""TEST INFRASTRUCTURE ONLY -- numpy front end of the C oracle (oracle/c/libgsa_oracle.so).

Drives the ``gsao_*`` entry points (same signatures as include/gsa.h, host pointers) through
the product's own ctypes table, so oracle and HIP library are called identically.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import os
import subprocess

import numpy as np

from gan_segmentation_amd import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIBRARY = os.path.join(_HERE, "c", "libgsa_oracle.so")
_api = None


def build(force=False):
    if force or not os.path.exists(ORACLE_LIBRARY) or (
            os.path.getmtime(ORACLE_LIBRARY) < os.path.getmtime(os.path.join(_HERE, "c", "gsa_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)


def api():
    global _api
    if _api is None:
        if not os.path.exists(ORACLE_LIBRARY):
            build()
        _api = _lib.Api(ORACLE_LIBRARY, "gsao_",
                        optional=("set_overlap", "segmentation_eval", "fill_inputs", "profile_enable", "profile_collect", "profile_entry", "profile_reset"))
    return _api


class Oracle:
    """Generator + decoder on host memory in the canonical fp32 order."""

    def __init__(self, gcfg=None, gparams=None, dcfg=None, dparams=None, precision="fp32"):
        self.ctx = _lib.Context(api(), 0)
        if precision != "fp32":
            self.ctx.set_precision(precision)      # operands of the MFMA convolutions rounded to bf16
        self.gcfg, self.dcfg = gcfg, dcfg
        if gcfg is not None:
            self.ctx.generator_init(gcfg)
            self.ctx.generator_load(gparams)
        if dcfg is not None:
            self.ctx.decoder_init(dcfg)
            self.ctx.decoder_load(dparams)

    def _gen_shapes(self, n):
        from gan_segmentation_amd.weights import generator_channels
        chans = generator_channels(self.gcfg)
        return [(n, c, 4 << i, 4 << i) for i, c in enumerate(chans)]

    def generator(self, z, noise, want_feats=True):
        """-> (rgb (N,3,R,R) f32, img (N,R,R,3) u8, [feats NCHW])"""
        z = np.ascontiguousarray(z, np.float32)
        noise = [np.ascontiguousarray(a, np.float32) for a in noise]
        n = z.shape[0]
        shapes = self._gen_shapes(n)
        R, nc = shapes[-1][2], self.gcfg["channels"]
        rgb = np.empty((n, nc, R, R), np.float32)
        img = np.empty((n, R, R, nc), np.uint8)
        feats = [np.empty(s, np.float32) for s in shapes] if want_feats else None
        self.ctx.generator_forward(None, n, z.ctypes.data, [a.ctypes.data for a in noise],
                                   rgb.ctypes.data, img.ctypes.data,
                                   [f.ctypes.data for f in feats] if want_feats else None)
        return rgb, img, feats

    def decoder(self, feats):
        """-> (logits (N,K,R,R) f32, mask (N,R,R) u8)"""
        feats = [np.ascontiguousarray(f, np.float32) for f in feats]
        n, R = feats[-1].shape[0], feats[-1].shape[2]
        k = self.dcfg["features"][-1]
        logits = np.empty((n, k, R, R), np.float32)
        mask = np.empty((n, R, R), np.uint8)
        self.ctx.decoder_forward(None, n, [f.ctypes.data for f in feats], logits.ctypes.data, mask.ctypes.data)
        return logits, mask

    def generate(self, z, noise):
        """-> (img (N,R,R,3) u8, mask (N,R,R) u8)"""
        z = np.ascontiguousarray(z, np.float32)
        noise = [np.ascontiguousarray(a, np.float32) for a in noise]
        n = z.shape[0]
        R, nc = 2 ** self.gcfg["max_res_log2"], self.gcfg["channels"]
        img = np.empty((n, R, R, nc), np.uint8)
        mask = np.empty((n, R, R), np.uint8)
        self.ctx.generate(None, n, z.ctypes.data, [a.ctypes.data for a in noise], img.ctypes.data, mask.ctypes.data)
        return img, mask
