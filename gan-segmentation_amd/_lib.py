"""ctypes binding of the C ABI declared in include/gsa.h.

``load_library()`` opens the HIP library built in-tree by ``__graft_entry__.build()``
(csrc/libgsa_hip.so).  There is no CPU fallback: if the library is missing or a call
fails, a ``GsaError`` is raised.  (The CPU oracle has a ctypes table of its own in
oracle/binding.py; this module never opens it.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIBRARY = os.path.join(_HERE, "csrc", "libgsa_hip.so")
EXPERIMENTS_LIBRARY = os.path.join(_HERE, "csrc", "libgsa_hip_exp.so")      # `make experiments`: + the measured-slower kernels behind their switches
if os.environ.get("GSA_HIP_LIBRARY"):      # tests / A-B runs: another build of the same library (a file name inside csrc/, or a path)
    _alt = os.environ["GSA_HIP_LIBRARY"]
    HIP_LIBRARY = _alt if os.path.isabs(_alt) else os.path.join(_HERE, "csrc", _alt)

# every symbol include/gsa.h declares
API_SYMBOLS = (
    "create", "destroy", "last_error", "generator_init", "generator_set_param",
    "generator_commit", "decoder_init", "decoder_set_param", "decoder_commit", "reserve",
    "generator_forward", "decoder_forward", "generate", "set_overlap", "set_precision", "segmentation_eval", "fill_inputs",
    "profile_enable", "profile_collect",
    "profile_entry", "profile_reset", "version", "check", "status_snapshot", "debug_inject",
)


class GsaError(RuntimeError):
    pass


PRECISIONS = {"fp32": 0, "bf16": 1}     # gsa_precision


class GeneratorConfig(ctypes.Structure):
    _fields_ = [("max_res_log2", ctypes.c_int32), ("fmap_base", ctypes.c_int32),
                ("fmap_decay", ctypes.c_double), ("fmap_max", ctypes.c_int32),
                ("latent_size", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("use_wscale", ctypes.c_int32)]


class DecoderConfig(ctypes.Structure):
    _fields_ = [("num_feats", ctypes.c_int32), ("start_res", ctypes.c_int32),
                ("use_bn", ctypes.c_int32), ("features", ctypes.POINTER(ctypes.c_int32)),
                ("in_channels", ctypes.POINTER(ctypes.c_int32))]


class Api:
    """Function table of one shared library exporting ``<prefix>create`` ... ."""

    def __init__(self, path, prefix="gsa_", optional=()):
        if not os.path.exists(path):
            raise GsaError("native library %s not found -- run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (there is no CPU fallback)" % path)
        self.path = path
        self.prefix = prefix
        if prefix == "gsa_":
            # The process must end up with ONE HIP runtime: torch brings its own libamdhip64, and a library that pulled
            # in another copy first leaves the GPU invisible to one of the two -- so torch is imported before the dlopen.
            import torch  # noqa: F401
        self.lib = ctypes.CDLL(path)
        c = ctypes
        vp, i32 = c.c_void_p, c.c_int32
        sig = {
            "create": (c.c_int, [c.c_int, c.POINTER(vp)]),
            "destroy": (None, [vp]),
            "last_error": (c.c_char_p, [vp]),
            "generator_init": (c.c_int, [vp, c.POINTER(GeneratorConfig)]),
            "generator_set_param": (c.c_int, [vp, c.c_char_p, vp, i32, c.POINTER(c.c_int64)]),
            "generator_commit": (c.c_int, [vp]),
            "decoder_init": (c.c_int, [vp, c.POINTER(DecoderConfig)]),
            "decoder_set_param": (c.c_int, [vp, c.c_char_p, vp, i32, c.POINTER(c.c_int64)]),
            "decoder_commit": (c.c_int, [vp]),
            "reserve": (c.c_int, [vp, i32]),
            "generator_forward": (c.c_int, [vp, vp, i32, vp, c.POINTER(vp), i32, vp, vp, c.POINTER(vp), i32]),
            "decoder_forward": (c.c_int, [vp, vp, i32, c.POINTER(vp), i32, vp, vp]),
            "generate": (c.c_int, [vp, vp, i32, vp, c.POINTER(vp), i32, vp, vp]),
            "set_overlap": (c.c_int, [vp, i32]),
            "set_precision": (c.c_int, [vp, i32]),
            "segmentation_eval": (c.c_int, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
            "fill_inputs": (c.c_int, [vp, vp, i32, c.c_uint64, c.c_uint64, vp, c.POINTER(vp), i32]),
            "profile_enable": (c.c_int, [vp, i32]),
            "profile_collect": (c.c_int, [vp]),
            "profile_entry": (c.c_int, [vp, i32, c.POINTER(c.c_char_p), c.POINTER(c.c_double),
                                        c.POINTER(c.c_int64), c.POINTER(c.c_double),
                                        c.POINTER(c.c_double), c.POINTER(c.c_double)]),
            "profile_reset": (c.c_int, [vp]),
            "version": (c.c_char_p, []),
            "check": (c.c_int, [vp]),
            "status_snapshot": (c.c_int, [vp, vp, vp]),
            "debug_inject": (c.c_int, [vp, i32, i32]),
        }
        for name, (res, args) in sig.items():
            try:
                fn = getattr(self.lib, prefix + name)
            except AttributeError:
                if name in optional:
                    continue
                raise GsaError("%s does not export %s%s" % (path, prefix, name))
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


_hip_api = None


def load_library():
    """The HIP library (cached).  Raises GsaError if it has not been built."""
    global _hip_api
    if _hip_api is None:
        _hip_api = Api(HIP_LIBRARY, "gsa_")
    return _hip_api


def _ptr_array(ptrs):
    arr = (ctypes.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p if p else None
    return arr


class Context:
    """One ``gsa_ctx``: a generator and/or decoder resident on one device."""

    def __init__(self, api, device=0):
        self.api = api
        self._h = ctypes.c_void_p()
        rc = api.create(int(device), ctypes.byref(self._h))
        if rc != 0:
            raise GsaError("create(device=%d) failed: %s" % (device, self._msg(None)))
        self.device = device
        self.generator_cfg = None
        self.decoder_cfg = None
        self.precision = "fp32"
        self._checked_first_step = False
        # hipGraph replay of small steps (image_generator._generate_on): graphs bake in the workspace pointers, the stream
        # structure (set_overlap) and must not contain profiling events -- any of these changing starts a new epoch
        self.graph_epoch = 0
        self.profiling = False

    def _msg(self, h):
        m = self.api.last_error(h)
        return m.decode("utf-8", "replace") if m else "unknown error"

    def _check(self, rc, what):
        if rc < 0:
            # any failing call ends the graph epoch: a pass that died half way leaves statistic rows that only the next EAGER
            # pass re-zeroes (gsa_api.cpp run_generator), and a failed commit / reserve leaves nothing a captured graph may use
            self.graph_epoch += 1
            raise GsaError("%s failed (%d): %s" % (what, rc, self._msg(self._h)))
        return rc

    def check(self):
        """gsa_check: synchronise the device and raise GsaError if a device-side check failed since the last clean one (the
        fused mapping network's exchange timed out; an instance-norm statistic left its fixed-point range)."""
        self._check(self.api.check(self._h), "check")

    def status_snapshot(self, stream, host_ptr):
        """gsa_status_snapshot: enqueue the 8-byte copy of the two sticky words to pinned host memory (no synchronisation)."""
        self._check(self.api.status_snapshot(self._h, stream, host_ptr), "status_snapshot")

    def debug_inject(self, kind, arg=0):
        """gsa_debug_inject (tests only)."""
        self._check(self.api.debug_inject(self._h, int(kind), int(arg)), "debug_inject")

    def _after_step(self):
        # ONE synchronising check per context, after its first step: a violated co-residency assumption or out-of-range
        # activations would otherwise yield wrong pairs with a zero status (the calls are stream-ordered); close() checks again
        if not self._checked_first_step:
            self._checked_first_step = True
            self.check()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            try:
                if self._checked_first_step:
                    self.check()
            finally:
                self.api.destroy(self._h)
                self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model setup ---------------------------------------------------------------
    def generator_init(self, cfg):
        gc = GeneratorConfig(int(cfg["max_res_log2"]), int(cfg["fmap_base"]), float(cfg["fmap_decay"]),
                             int(cfg["fmap_max"]), int(cfg["latent_size"]), int(cfg["channels"]),
                             1 if cfg["use_wscale"] else 0)
        if int(cfg.get("base_scale_x", 4)) != 4 or int(cfg.get("base_scale_y", 4)) != 4:
            raise GsaError("only the 4x4 base resolution of the reference config is supported")
        self._check(self.api.generator_init(self._h, ctypes.byref(gc)), "generator_init")
        self.generator_cfg = dict(cfg)

    def _set_params(self, fn, params, what):
        ignored = []
        for name, arr in params.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            dims = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            rc = self._check(fn(self._h, name.encode("utf-8"), a.ctypes.data, a.ndim, dims),
                             "%s(%s)" % (what, name))
            if rc == 1:
                ignored.append(name)
        return ignored

    def generator_load(self, params):
        ignored = self._set_params(self.api.generator_set_param, params, "generator_set_param")
        self.graph_epoch += 1       # the commit frees and re-uploads the weight panels a captured graph bakes in
        self._check(self.api.generator_commit(self._h), "generator_commit")
        return ignored

    def decoder_init(self, cfg):
        feats = (ctypes.c_int32 * len(cfg["features"]))(*cfg["features"])
        inch = (ctypes.c_int32 * len(cfg["in_channels"]))(*cfg["in_channels"])
        if len(cfg["features"]) != len(cfg["in_channels"]) + 1:
            raise GsaError("decoder cfg: len(features) must be len(in_channels)+1")
        dc = DecoderConfig(len(cfg["in_channels"]), int(cfg["start_res"]), 1 if cfg["use_bn"] else 0,
                           feats, inch)
        self._check(self.api.decoder_init(self._h, ctypes.byref(dc)), "decoder_init")
        self.decoder_cfg = dict(cfg)

    def decoder_load(self, params):
        self._set_params(self.api.decoder_set_param, params, "decoder_set_param")
        self.graph_epoch += 1
        self._check(self.api.decoder_commit(self._h), "decoder_commit")

    def reserve(self, max_batch):
        self.graph_epoch += 1
        self._check(self.api.reserve(self._h, int(max_batch)), "reserve")

    # -- forward calls: every tensor argument is a raw address (int) or None -------------
    def generator_forward(self, stream, n, z, noise, rgb=None, img=None, feats=None):
        fp = _ptr_array(feats) if feats is not None else None
        self._check(self.api.generator_forward(self._h, stream, n, z, _ptr_array(noise), len(noise), rgb, img, fp,
                                               len(feats) if feats is not None else 0), "generator_forward")
        self._after_step()

    def decoder_forward(self, stream, n, feats, logits=None, mask=None):
        self._check(self.api.decoder_forward(self._h, stream, n, _ptr_array(feats), len(feats), logits, mask),
                    "decoder_forward")

    def generate(self, stream, n, z, noise, img, mask):
        self._check(self.api.generate(self._h, stream, n, z, _ptr_array(noise), len(noise), img, mask), "generate")
        self._after_step()

    def set_precision(self, precision):
        """"fp32" (default, bit-exact canonical path) or "bf16" (bf16 MFMA operands); before the weights are loaded."""
        self._check(self.api.set_precision(self._h, PRECISIONS[precision]), "set_precision")
        self.precision = precision

    def fill_inputs(self, stream, n, seed, first_index, z=None, noise=None):
        self._check(self.api.fill_inputs(self._h, stream, n, int(seed) & (2 ** 64 - 1), int(first_index), z,
                                         _ptr_array(noise) if noise is not None else None,
                                         len(noise) if noise is not None else 0), "fill_inputs")

    def segmentation_eval(self, stream, n, classes, H, W, logits, labels, confusion, loss_fixed):
        self._check(self.api.segmentation_eval(self._h, stream, n, classes, H, W, logits, labels, confusion, loss_fixed),
                    "segmentation_eval")

    def set_overlap(self, levels):
        self.graph_epoch += 1
        self._check(self.api.set_overlap(self._h, int(levels)), "set_overlap")

    # -- measurement ----------------------------------------------------------------------
    def profile_enable(self, on=True):
        self.graph_epoch += 1
        self.profiling = bool(int(on))
        self._check(self.api.profile_enable(self._h, int(on)), "profile_enable")

    def profile_reset(self):
        self._check(self.api.profile_reset(self._h), "profile_reset")

    def profile_entries(self):
        n = self._check(self.api.profile_collect(self._h), "profile_collect")
        out = []
        for i in range(n):
            name = ctypes.c_char_p()
            ms, fl, by, af = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            cnt = ctypes.c_int64()
            self._check(self.api.profile_entry(self._h, i, ctypes.byref(name), ctypes.byref(ms),
                                               ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(af)),
                        "profile_entry")
            out.append({"name": name.value.decode(), "ms": ms.value, "launches": cnt.value,
                        "flops": fl.value, "bytes": by.value, "alg_flops": af.value})
        return out
