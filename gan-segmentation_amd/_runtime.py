"""Torch interop for the HIP library: PyTorch is used for device memory, streams and
``torch.distributed`` only -- every computation goes through the C ABI (include/gsa.h)."""
import numpy as np
import torch

from . import _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.GsaError("no HIP device visible to PyTorch: the generate path has no CPU fallback")


def current_stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def to_device_f32(x, device):
    """numpy / torch (any device) -> contiguous fp32 torch tensor on ``device``."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x, dtype=np.float32))
    return x.to(device=device, dtype=torch.float32, non_blocking=True).contiguous()


class DeviceModel:
    """ONE ``gsa_ctx`` on one torch device with one arithmetic ("fp32" / "bf16" MFMA operands).

    Every ``Generator`` and every stand-alone ``Decoder`` owns its own DeviceModel, as the reference's gluon
    blocks own their parameters (two generators in one process never see each other's weights or
    configuration).  A decoder that is to run fused with a generator (``ImageGenerator.attach_decoder``) is
    loaded into THAT generator's context, because ``gsa_generate`` hands the features over inside it."""

    def __init__(self, device_index, precision="fp32"):
        require_gpu()
        if precision not in _lib.PRECISIONS:
            raise _lib.GsaError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        if not 0 <= int(device_index) < torch.cuda.device_count():
            raise _lib.GsaError("gpu id %r: this process sees %d HIP device(s)" % (device_index, torch.cuda.device_count()))
        self.device = torch.device("cuda", int(device_index))
        self.precision = precision
        self.ctx = _lib.Context(_lib.load_library(), int(device_index))
        if precision != "fp32":
            self.ctx.set_precision(precision)
        self.reserved = 0
        self.generator_cfg = None
        self.decoder_cfg = None

    def ensure_batch(self, n):
        if n > self.reserved:
            torch.cuda.synchronize(self.device)
            self.ctx.reserve(n)
            self.reserved = n

    def invalidate_workspace(self):
        self.reserved = 0


def split_sizes(total, parts):
    """Sizes of the contiguous slices a batch of ``total`` samples is cut into for ``parts`` devices
    (``gluon.utils.split_and_load(even_split=False)``, reference image_generator.py:95): empty slices are dropped."""
    from .dist import shard_bounds
    out = []
    for r in range(parts):
        lo, hi = shard_bounds(total, parts, r)
        if hi > lo:
            out.append((r, lo, hi))
    return out
