"""``Decoder`` with the call surface of reference networks_seg.py:49-113 (inference only),
executed by the HIP library.

    net = Decoder(cfg, num_devices)         # reference networks_seg.py:51-94
    net.load_parameters(path)               # reference seg_solver.py:339-349
    logits = net(*features)                 # reference networks_seg.py:97-113
"""
import torch

from . import params as _params
from . import weights as _weights
from ._runtime import DeviceModel, current_stream_ptr, to_device_f32


class Decoder:
    def __init__(self, cfg, num_devices=1, device=0, precision="fp32", model=None, **kwargs):
        self.cfg = dict(cfg)
        for key in ("features", "in_channels", "start_res", "use_bn", "use_sync_bn", "use_dropout"):
            if key not in self.cfg:
                raise KeyError("Decoder cfg is missing %r" % key)  # reference :54-62
        self._num_devices = num_devices
        self.precision = precision
        # its own context unless it is to run fused with a generator (``model`` = that generator's context)
        self._model = model if model is not None else DeviceModel(device, precision)
        if self._model.decoder_cfg is not None:
            raise RuntimeError("this context already holds a Decoder")
        self.precision = self._model.precision
        self._model.ctx.decoder_init(self.cfg)
        self._model.decoder_cfg = self.cfg
        self._model.invalidate_workspace()
        self._loaded = False
        self._tensors = None

    def hybridize(self, *args, **kwargs):  # reference seg_solver.py:41 -- nothing to trace here
        return None

    def load_parameters(self, source, ctx=None, **kwargs):
        tensors = _params.load_params(source) if isinstance(source, (str, bytes)) else dict(source)
        tensors = _weights.complete_decoder_params(self.cfg, tensors)
        self._tensors = tensors
        self._model.ctx.decoder_load(tensors)
        self._model.invalidate_workspace()
        self._loaded = True

    def save_parameters(self, path):
        """Write the structural-name ``.params`` file the reference writes (seg_solver.py:331-337)."""
        if not self._loaded:
            raise RuntimeError("Decoder parameters are not loaded")
        _params.save_params(path, self._tensors)

    def _prepare(self, features):
        if not self._loaded:
            raise RuntimeError("Decoder parameters are not loaded")
        dev = self._model.device
        feats = [to_device_f32(f, dev) for f in features]
        feats = [f.unsqueeze(0) if f.dim() == 3 else f for f in feats]
        inch = self.cfg["in_channels"]
        if len(feats) != len(inch):
            raise ValueError("expected %d feature maps, got %d" % (len(inch), len(feats)))
        n = feats[0].shape[0]
        for i, f in enumerate(feats):
            if tuple(f.shape) != (n, inch[i], 4 << i, 4 << i):
                raise ValueError("feature %d has shape %s, expected %s" % (i, tuple(f.shape), (n, inch[i], 4 << i, 4 << i)))
        self._model.ensure_batch(n)
        return feats, n

    def __call__(self, *features, want_mask=False):
        feats, n = self._prepare(features)
        dev = self._model.device
        R = 4 << (len(feats) - 1)
        k = self.cfg["features"][-1]
        logits = torch.empty((n, k, R, R), device=dev, dtype=torch.float32)
        mask = torch.empty((n, R, R), device=dev, dtype=torch.uint8) if want_mask else None
        self._model.ctx.decoder_forward(current_stream_ptr(dev), n, [f.data_ptr() for f in feats], logits.data_ptr(),
                                        mask.data_ptr() if want_mask else None)
        return (logits, mask) if want_mask else logits

    forward = __call__
