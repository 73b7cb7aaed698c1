"""TEST INFRASTRUCTURE ONLY -- numpy front end of the C oracle (oracle/c/libgsa_oracle.so).

Drives the ``gsao_*`` entry points (the signatures of include/gsa.h with host pointers) through its OWN ctypes table:
nothing of the product package is imported here, so a marshalling mistake in ``gan_segmentation_amd/_lib.py`` cannot
cancel out between the oracle and the HIP library (round 1 shared that module).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIBRARY = os.path.join(_HERE, "c", "libgsa_oracle.so")
_lib = None


class OracleError(RuntimeError):
    pass


class _GenCfg(ctypes.Structure):        # gsa_generator_config (include/gsa.h)
    _fields_ = [("max_res_log2", ctypes.c_int32), ("fmap_base", ctypes.c_int32), ("fmap_decay", ctypes.c_double),
                ("fmap_max", ctypes.c_int32), ("latent_size", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("use_wscale", ctypes.c_int32)]


class _DecCfg(ctypes.Structure):        # gsa_decoder_config
    _fields_ = [("num_feats", ctypes.c_int32), ("start_res", ctypes.c_int32), ("use_bn", ctypes.c_int32),
                ("features", ctypes.POINTER(ctypes.c_int32)), ("in_channels", ctypes.POINTER(ctypes.c_int32))]


def build(force=False):
    if force or not os.path.exists(ORACLE_LIBRARY) or (
            os.path.getmtime(ORACLE_LIBRARY) < os.path.getmtime(os.path.join(_HERE, "c", "gsa_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)


def api():
    """The oracle library with argument types declared (cached)."""
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIBRARY):
            build()
        lib = ctypes.CDLL(ORACLE_LIBRARY)
        c = ctypes
        vp, i32, pp = c.c_void_p, c.c_int32, c.POINTER(c.c_void_p)
        sig = {
            "gsao_create": (c.c_int, [c.c_int, pp]),
            "gsao_destroy": (None, [vp]),
            "gsao_last_error": (c.c_char_p, [vp]),
            "gsao_set_precision": (c.c_int, [vp, i32]),
            "gsao_generator_init": (c.c_int, [vp, c.POINTER(_GenCfg)]),
            "gsao_generator_set_param": (c.c_int, [vp, c.c_char_p, vp, i32, c.POINTER(c.c_int64)]),
            "gsao_generator_commit": (c.c_int, [vp]),
            "gsao_decoder_init": (c.c_int, [vp, c.POINTER(_DecCfg)]),
            "gsao_decoder_set_param": (c.c_int, [vp, c.c_char_p, vp, i32, c.POINTER(c.c_int64)]),
            "gsao_decoder_commit": (c.c_int, [vp]),
            "gsao_generator_forward": (c.c_int, [vp, vp, i32, vp, pp, i32, vp, vp, pp, i32]),
            "gsao_decoder_forward": (c.c_int, [vp, vp, i32, pp, i32, vp, vp]),
            "gsao_generate": (c.c_int, [vp, vp, i32, vp, pp, i32, vp, vp]),
            "gsao_version": (c.c_char_p, []),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _ptrs(arrays):
    arr = (ctypes.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ctypes.data if a is not None else None
    return arr


class Oracle:
    """Generator + decoder on host memory in the canonical fp32 order (``precision="bf16"``: the bf16 mode's roundings)."""

    def __init__(self, gcfg=None, gparams=None, dcfg=None, dparams=None, precision="fp32"):
        self.lib = api()
        self._h = ctypes.c_void_p()
        if self.lib.gsao_create(0, ctypes.byref(self._h)) != 0:
            raise OracleError("gsao_create failed")
        if precision != "fp32":
            self._check(self.lib.gsao_set_precision(self._h, {"bf16": 1}[precision]), "set_precision")
        self.gcfg, self.dcfg = gcfg, dcfg
        if gcfg is not None:
            cfg = _GenCfg(int(gcfg["max_res_log2"]), int(gcfg["fmap_base"]), float(gcfg["fmap_decay"]), int(gcfg["fmap_max"]),
                          int(gcfg["latent_size"]), int(gcfg["channels"]), 1 if gcfg["use_wscale"] else 0)
            self._check(self.lib.gsao_generator_init(self._h, ctypes.byref(cfg)), "generator_init")
            self._load(self.lib.gsao_generator_set_param, gparams)
            self._check(self.lib.gsao_generator_commit(self._h), "generator_commit")
        if dcfg is not None:
            feats = (ctypes.c_int32 * len(dcfg["features"]))(*dcfg["features"])
            inch = (ctypes.c_int32 * len(dcfg["in_channels"]))(*dcfg["in_channels"])
            cfg = _DecCfg(len(dcfg["in_channels"]), int(dcfg["start_res"]), 1 if dcfg["use_bn"] else 0, feats, inch)
            self._check(self.lib.gsao_decoder_init(self._h, ctypes.byref(cfg)), "decoder_init")
            self._load(self.lib.gsao_decoder_set_param, dparams)
            self._check(self.lib.gsao_decoder_commit(self._h), "decoder_commit")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.lib.gsao_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            msg = self.lib.gsao_last_error(self._h)
            raise OracleError("%s failed (%d): %s" % (what, rc, msg.decode("utf-8", "replace") if msg else "?"))
        return rc

    def _load(self, fn, params):
        for name, arr in params.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            dims = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            self._check(fn(self._h, name.encode("utf-8"), a.ctypes.data, a.ndim, dims), "set_param(%s)" % name)

    def _gen_shapes(self, n):
        g = self.gcfg
        chans = [min(int(g["fmap_base"] / (2.0 ** ((r - 1) * g["fmap_decay"]))), g["fmap_max"])      # networks_stylegan.py:114-116
                 for r in range(2, g["max_res_log2"] + 1)]
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
        self._check(self.lib.gsao_generator_forward(self._h, None, n, z.ctypes.data, _ptrs(noise), len(noise), rgb.ctypes.data,
                                                    img.ctypes.data, _ptrs(feats) if want_feats else None,
                                                    len(feats) if want_feats else 0), "generator_forward")
        return rgb, img, feats

    def decoder(self, feats):
        """-> (logits (N,K,R,R) f32, mask (N,R,R) u8)"""
        feats = [np.ascontiguousarray(f, np.float32) for f in feats]
        n, R = feats[-1].shape[0], feats[-1].shape[2]
        k = self.dcfg["features"][-1]
        logits = np.empty((n, k, R, R), np.float32)
        mask = np.empty((n, R, R), np.uint8)
        self._check(self.lib.gsao_decoder_forward(self._h, None, n, _ptrs(feats), len(feats), logits.ctypes.data, mask.ctypes.data),
                    "decoder_forward")
        return logits, mask

    def generate(self, z, noise):
        """-> (img (N,R,R,3) u8, mask (N,R,R) u8)"""
        z = np.ascontiguousarray(z, np.float32)
        noise = [np.ascontiguousarray(a, np.float32) for a in noise]
        n = z.shape[0]
        R, nc = 2 ** self.gcfg["max_res_log2"], self.gcfg["channels"]
        img = np.empty((n, R, R, nc), np.uint8)
        mask = np.empty((n, R, R), np.uint8)
        self._check(self.lib.gsao_generate(self._h, None, n, z.ctypes.data, _ptrs(noise), len(noise), img.ctypes.data, mask.ctypes.data),
                    "generate")
        return img, mask
