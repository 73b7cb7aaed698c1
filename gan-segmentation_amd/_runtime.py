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
    """A ``gsa_ctx`` bound to one torch device and one arithmetic ("fp32" / "bf16" MFMA operands), shared by
    Generator and Decoder objects that live on the same GPU (so the fused ``generate`` call sees both)."""

    _by_device = {}

    def __init__(self, device_index, precision="fp32"):
        require_gpu()
        if precision not in _lib.PRECISIONS:
            raise _lib.GsaError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        self.device = torch.device("cuda", device_index)
        self.precision = precision
        self.ctx = _lib.Context(_lib.load_library(), device_index)
        if precision != "fp32":
            self.ctx.set_precision(precision)
        self.reserved = 0
        self.generator_cfg = None
        self.decoder_cfg = None

    @classmethod
    def get(cls, device_index=0, precision="fp32"):
        m = cls._by_device.get((device_index, precision))
        if m is None:
            m = cls(device_index, precision)
            cls._by_device[(device_index, precision)] = m
        return m

    def ensure_batch(self, n):
        if n > self.reserved:
            torch.cuda.synchronize(self.device)
            self.ctx.reserve(n)
            self.reserved = n

    def invalidate_workspace(self):
        self.reserved = 0
