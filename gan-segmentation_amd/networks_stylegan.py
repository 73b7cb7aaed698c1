"""``Generator`` with the call surface of reference networks_stylegan.py:76-197, executed by
the HIP library.

    netG = Generator(config)                       # reference networks_stylegan.py:78-112
    netG.load_parameters(path, ignore_extra=True)  # reference image_generator.py:21-22
    rgb, features = netG(z)                        # reference networks_stylegan.py:165-197

``rgb`` is (N,3,R,R) fp32 and ``features`` the list of (N,C_r,R_r,R_r) fp32 block outputs,
as torch tensors on the GPU (the analogue of the reference's device NDArrays).

Differences from the reference, all additive: the AddNoise planes, which the reference
draws from MXNet's global RNG inside the network (networks_stylegan.py:297-300), can be
passed explicitly (``noise=[...]``) so that results are reproducible; without them they are
drawn on the device from ``torch.Generator``.
"""
import numpy as np
import torch

from . import params as _params
from . import weights as _weights
from ._runtime import DeviceModel, current_stream_ptr, to_device_f32


class Generator:
    def __init__(self, config, device=0, precision="fp32", model=None, **kwargs):
        self.config = dict(config)
        for key in ("fmap_base", "fmap_decay", "fmap_max", "base_scale_x", "base_scale_y", "use_wscale",
                    "fix_noise", "channels", "latent_size", "max_res_log2"):
            if key not in self.config:
                raise KeyError("Generator config is missing %r" % key)  # reference :81-91
        self.fix_noise = bool(self.config["fix_noise"])
        self.nc = self.config["channels"]
        self.latent_size = self.config["latent_size"]
        self.max_res_log2 = self.config["max_res_log2"]
        self.precision = precision
        # its own context: a second Generator on the same GPU must not touch this one's weights or configuration
        self._model = model if model is not None else DeviceModel(device, precision)
        if self._model.generator_cfg is not None:
            raise RuntimeError("this context already holds a Generator")
        self._model.ctx.generator_init(self.config)
        self._model.generator_cfg = self.config
        self._model.invalidate_workspace()
        self._loaded = False
        self._fixed_noise = None
        self._rng = torch.Generator(device=self._model.device)
        self._rng.manual_seed(0)

    # -- parameters -----------------------------------------------------------------------
    def num_features(self, res_log2):
        return _weights.num_features(self.config, res_log2)

    def load_parameters(self, source, ignore_extra=True, ctx=None, allow_missing=False):
        """``source``: a ``.params`` path or a ``{name: ndarray}`` dict (either naming scheme)."""
        if allow_missing:
            raise NotImplementedError("allow_missing is not supported (the reference does not use it)")
        tensors = _params.load_params(source) if isinstance(source, (str, bytes)) else dict(source)
        tensors = _weights.complete_generator_params(self.config, tensors, ignore_extra=ignore_extra)
        self._model.ctx.generator_load(tensors)
        self._model.invalidate_workspace()
        self._loaded = True

    def seed(self, seed):
        self._rng.manual_seed(int(seed))

    # -- forward --------------------------------------------------------------------------
    def noise_shapes(self, batch):
        return [(batch, 1, 2 ** r, 2 ** r) for r in range(2, self.max_res_log2 + 1) for _ in range(2)]

    def draw_noise(self, batch):
        dev = self._model.device
        if self.fix_noise and self._fixed_noise is not None and self._fixed_noise[0].shape[0] == batch:
            return self._fixed_noise
        noise = [torch.randn(s, device=dev, dtype=torch.float32, generator=self._rng) for s in self.noise_shapes(batch)]
        if self.fix_noise:
            self._fixed_noise = noise
        return noise

    def draw_indexed(self, first_index, n, seed=0):
        """(z (n,latent), [noise planes]) on the device as a pure function of (seed, global sample index): sample
        ``first_index + k`` gets the same bytes whatever batch, rank or GPU count asks for it (``gsa_fill_inputs``,
        Philox4x32-10 + Box-Muller)."""
        if not self._loaded:
            raise RuntimeError("Generator parameters are not loaded")
        dev = self._model.device
        z = torch.empty((n, self.latent_size), device=dev, dtype=torch.float32)
        noise = [torch.empty(s, device=dev, dtype=torch.float32) for s in self.noise_shapes(n)]
        self._model.ctx.fill_inputs(current_stream_ptr(dev), n, seed, first_index, z.data_ptr(), [a.data_ptr() for a in noise])
        return z, noise

    def _prepare(self, z, noise):
        if not self._loaded:
            raise RuntimeError("Generator parameters are not loaded")
        dev = self._model.device
        z = to_device_f32(z, dev)
        if z.dim() != 2 or z.shape[1] != self.latent_size:
            raise ValueError("z must have shape (N, %d)" % self.latent_size)
        n = z.shape[0]
        if noise is None:
            noise = self.draw_noise(n)
        else:
            noise = [to_device_f32(a, dev) for a in noise]
            shapes = self.noise_shapes(n)
            if len(noise) != len(shapes) or any(tuple(a.shape) != s for a, s in zip(noise, shapes)):
                raise ValueError("noise must be %d planes shaped (N,1,R,R), R=4,4,8,8,..." % len(shapes))
        self._model.ensure_batch(n)
        return z, noise, n

    def __call__(self, z, noise=None, want_features=True, want_image=False):
        z, noise, n = self._prepare(z, noise)
        dev = self._model.device
        R = 2 ** self.max_res_log2
        rgb = torch.empty((n, self.nc, R, R), device=dev, dtype=torch.float32)
        img = torch.empty((n, R, R, self.nc), device=dev, dtype=torch.uint8) if want_image else None
        feats = None
        if want_features:
            chans = _weights.generator_channels(self.config)
            feats = [torch.empty((n, c, 4 << i, 4 << i), device=dev, dtype=torch.float32) for i, c in enumerate(chans)]
        self._model.ctx.generator_forward(
            current_stream_ptr(dev), n, z.data_ptr(), [a.data_ptr() for a in noise], rgb.data_ptr(),
            img.data_ptr() if img is not None else None, [f.data_ptr() for f in feats] if feats else None)
        if want_image:
            return rgb, feats, img
        return rgb, feats

    forward = __call__


# The north-star wording calls the synthesis part "SynthesisNetwork"; the reference has one class.
SynthesisNetwork = Generator
