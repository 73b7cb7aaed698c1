"""``ImageGenerator`` with the call surface of reference image_generator.py:6-124.

    netG = ImageGenerator(gpu_ids, gan_dir, gan='ffhq', batch_size=4, return_latents=False)
    for img, feats in netG.get_images(n): ...       # img (R,R,3) u8 RGB, feats [C,R_r,R_r] fp32

Host-visible behaviour follows the reference: batches of ``batch_size`` latents, the last one
short (``:88-92``), per-sample yield of ``(img, feats[, latent_z_np])`` where ``latent_z_np`` is
the WHOLE batch array (``:105,122``).  Additions: explicit latents/noise for reproducibility,
``keep_on_device`` to skip the 132.8 MB/sample host round trip of the reference
(``:103-114``), and ``generate_batch`` -- the fused generator+decoder step used by
``main.py generate`` -- which never materialises the features.
"""
import os

import numpy as np
import torch

from . import weights as _weights
from ._runtime import DeviceModel, current_stream_ptr, to_device_f32
from .networks_seg import Decoder
from .networks_stylegan import Generator


class ImageGenerator:
    def __init__(self, gpu_ids, gan_dir, gan="ffhq", batch_size=4, return_latents=False, seed=0, precision="fp32"):
        max_res_log2_dict = _weights.GAN_MAX_RES_LOG2
        self.max_res_log2 = max_res_log2_dict[gan]
        self.latent_size = 512
        self.return_latents = return_latents
        self.batch_size = batch_size
        gpu_ids = list(gpu_ids)
        if len(gpu_ids) == 0:
            raise RuntimeError("the MI355X path has no CPU context: pass at least one gpu id "
                               "(the reference falls back to mx.cpu(), image_generator.py:17)")
        if len(gpu_ids) > 1:
            raise RuntimeError("one process drives one GPU; shard batches across ranks with "
                               "gan_segmentation_amd.dist (one process per GPU over RCCL)")
        self.ctx = gpu_ids
        self.precision = precision
        self.cfg = self._get_config(max_res_log2=self.max_res_log2)
        self.netG = self._get_G(self.cfg, gpu_ids[0])
        stylegan_name = "stylegan-%s.params" % gan
        self.netG.load_parameters(os.path.join(gan_dir, stylegan_name), ignore_extra=True)
        self._decoder = None
        self._rng = torch.Generator(device="cpu")
        self._rng.manual_seed(seed)
        self.netG.seed(seed)

    @classmethod
    def from_params(cls, gcfg, gparams, dcfg=None, dparams=None, gpu_ids=(0,), batch_size=4,
                    return_latents=False, seed=0, precision="fp32"):
        """Build from in-memory weights (tests, benchmarks: no pretrained files exist here)."""
        self = cls.__new__(cls)
        self.max_res_log2 = gcfg["max_res_log2"]
        self.latent_size = gcfg["latent_size"]
        self.return_latents = return_latents
        self.batch_size = batch_size
        self.ctx = list(gpu_ids)
        self.cfg = dict(gcfg)
        self.precision = precision
        self.netG = Generator(self.cfg, device=self.ctx[0], precision=precision)
        self.netG.load_parameters(gparams)
        self._decoder = None
        if dcfg is not None:
            self.attach_decoder(dcfg, dparams)
        self._rng = torch.Generator(device="cpu")
        self._rng.manual_seed(seed)
        self.netG.seed(seed)
        return self

    def _get_G(self, config, device):
        return Generator(config, device=device, precision=self.precision)

    def _get_config(self, max_res_log2=9):
        return _weights.generator_config(max_res_log2)  # reference image_generator.py:46-74

    def attach_decoder(self, dcfg, dparams):
        """Put a decoder on the same GPU so ``generate_batch`` can run the fused path."""
        dec = dparams if isinstance(dparams, Decoder) else None
        if dec is None:
            dec = Decoder(dcfg, 1, device=self.ctx[0], precision=self.precision)
            dec.load_parameters(dparams)
        if dec._model is not self.netG._model:
            raise RuntimeError("the decoder lives on another device or precision than the generator "
                               "(%s vs %s)" % (dec.precision, self.precision))
        self._decoder = dec
        return dec

    @staticmethod
    def _transform_gan_back(img, cfg):
        """numpy restatement of reference image_generator.py:76-84 (kept for callers that hold
        fp32 rgb; the device path produces the same bytes in toRGB's epilogue)."""
        lo, hi = cfg["imrange"]
        img = np.transpose(img, (0, 2, 3, 1))
        img = (img - np.float32(lo)) / np.float32(hi - lo)
        img = np.clip(img, 0.0, 1.0)
        return (np.float32(255.0) * img).astype(np.uint8)

    def draw_latents(self, n):
        return torch.randn((n, self.latent_size), generator=self._rng, dtype=torch.float32)

    # -- reference surface ------------------------------------------------------------------
    def get_images(self, n, latents=None, noise=None, keep_on_device=False):
        n_batches = n // self.batch_size + (1 if n % self.batch_size > 0 else 0)
        n_generated = 0
        for _ in range(n_batches):
            bs = min(self.batch_size, n - n_generated)
            if latents is not None:
                latent_z = torch.as_tensor(np.asarray(latents[n_generated:n_generated + bs], dtype=np.float32))
            else:
                latent_z = self.draw_latents(bs)
            nz = None
            if noise is not None:
                nz = [a[n_generated:n_generated + bs] for a in noise]
            _rgb, feats, imgs = self.netG(latent_z, noise=nz, want_image=True)
            latent_z_np = latent_z.numpy()
            if not keep_on_device:
                torch.cuda.synchronize()
                imgs = imgs.cpu().numpy()
                feats = [f.cpu().numpy() for f in feats]
            n_generated += bs
            for i in range(bs):
                img = imgs[i]
                fs = [f[i] for f in feats]
                if self.return_latents:
                    yield img, fs, latent_z_np
                else:
                    yield img, fs

    # -- fused hot path ---------------------------------------------------------------------
    def generate_indexed(self, first_index, n, seed=0, out=None):
        """``generate_batch`` for the global samples ``first_index .. first_index+n-1`` with counter-based latents
        and noise (``Generator.draw_indexed``): the dataset does not depend on how it is sharded."""
        z, noise = self.netG.draw_indexed(first_index, n, seed)
        return self.generate_batch(z, noise, out=out)

    def generate_batch(self, z, noise=None, out=None):
        """latents (N,512) [+ noise planes] -> (img (N,R,R,3) u8, mask (N,R,R) u8) on the GPU.
        The per-batch body of ``main.py generate`` (reference main.py:97-99) in one call.
        ``out=(img, mask)``: write into these contiguous uint8 device tensors instead of new ones
        (e.g. the fused send buffer of ``dist.PairGatherer``)."""
        if self._decoder is None:
            raise RuntimeError("attach_decoder() first")
        g = self.netG
        z, noise, n = g._prepare(z, noise)
        model = g._model
        dev = model.device
        R = 2 ** self.max_res_log2
        if out is None:
            img = torch.empty((n, R, R, g.nc), device=dev, dtype=torch.uint8)
            mask = torch.empty((n, R, R), device=dev, dtype=torch.uint8)
        else:
            img, mask = out
            for t, shape in ((img, (n, R, R, g.nc)), (mask, (n, R, R))):
                if tuple(t.shape) != shape or t.dtype != torch.uint8 or t.device != dev or not t.is_contiguous():
                    raise ValueError("out tensors must be contiguous uint8 %s on %s" % (shape, dev))
        model.ctx.generate(current_stream_ptr(dev), n, z.data_ptr(), [a.data_ptr() for a in noise],
                           img.data_ptr(), mask.data_ptr())
        return img, mask
