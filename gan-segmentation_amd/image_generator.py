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
from ._runtime import current_stream_ptr, split_sizes, to_device_f32
from .dataset_writer import STATUS_RING_DEPTH
from .networks_seg import Decoder
from .networks_stylegan import Generator


_GRAPH_CAPTURES_MAX = 16


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
        self.ctx = gpu_ids      # the reference's device list (image_generator.py:17): one weight replica per entry
        self.precision = precision
        self.cfg = self._get_config(max_res_log2=self.max_res_log2)
        stylegan_name = "stylegan-%s.params" % gan
        from . import params as _params
        tensors = _params.load_params(os.path.join(gan_dir, stylegan_name))     # read once, loaded into every replica
        self._gens = []
        for dev in gpu_ids:
            g = self._get_G(self.cfg, dev)
            g.load_parameters(tensors, ignore_extra=True)
            self._gens.append(g)
        self.netG = self._gens[0]
        self._decoder = None
        self._decoders = []
        self._rng = torch.Generator(device="cpu")
        self._rng.manual_seed(seed)
        for g in self._gens:
            g.seed(seed)

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
        self._gens = []
        for dev in self.ctx:
            g = Generator(self.cfg, device=dev, precision=precision)
            g.load_parameters(gparams)
            self._gens.append(g)
        self.netG = self._gens[0]
        self._decoder = None
        self._decoders = []
        if dcfg is not None:
            self.attach_decoder(dcfg, dparams)
        self._rng = torch.Generator(device="cpu")
        self._rng.manual_seed(seed)
        for g in self._gens:
            g.seed(seed)
        return self

    def _get_G(self, config, device):
        return Generator(config, device=device, precision=self.precision)

    def _get_config(self, max_res_log2=9):
        return _weights.generator_config(max_res_log2)  # reference image_generator.py:46-74

    def attach_decoder(self, dcfg, dparams):
        """Load a decoder into the context of every generator replica so that ``generate_batch`` can run the fused
        path (the features never leave the kernels' layout).  ``dparams``: a ``{name: array}`` dict, a ``.params``
        path, or a loaded ``Decoder`` (e.g. ``SegSolver.net``) whose weights are copied -- that object stays
        independent, as two gluon blocks would."""
        if isinstance(dparams, Decoder):
            if dparams._tensors is None:
                raise RuntimeError("the decoder to attach has no parameters loaded")
            dcfg, dparams = dparams.cfg, dparams._tensors
        self._decoders = []
        for g in self._gens:
            if g._model.decoder_cfg is not None:        # re-attach: a fresh decoder slot in the same context
                g._model.decoder_cfg = None
            dec = Decoder(dcfg, len(self._gens), model=g._model)
            dec.load_parameters(dparams)
            self._decoders.append(dec)
        self._decoder = self._decoders[0]
        return self._decoder

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
            # data-parallel split over the device list, host-side gather (reference :95-114); the launches of all
            # devices are enqueued before the first result is awaited
            parts = []
            for r, lo, hi in split_sizes(bs, len(self._gens)):
                g = self._gens[r]
                with torch.cuda.device(g._model.device):
                    parts.append(g(latent_z[lo:hi], noise=None if nz is None else [a[lo:hi] for a in nz], want_image=True))
            latent_z_np = latent_z.numpy()
            if not keep_on_device:
                for g in self._gens:
                    torch.cuda.synchronize(g._model.device)
                imgs = np.concatenate([p[2].cpu().numpy() for p in parts], axis=0)
                feats = [np.concatenate([p[1][i].cpu().numpy() for p in parts], axis=0) for i in range(len(parts[0][1]))]
            else:
                dev0 = self._gens[0]._model.device
                imgs = torch.cat([p[2].to(dev0) for p in parts], dim=0) if len(parts) > 1 else parts[0][2]
                feats = ([torch.cat([p[1][i].to(dev0) for p in parts], dim=0) for i in range(len(parts[0][1]))]
                         if len(parts) > 1 else parts[0][1])
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
        if len(self._gens) == 1:
            z, noise = self.netG.draw_indexed(first_index, n, seed)
            return self.generate_batch(z, noise, out=out)
        parts = []
        for r, lo, hi in split_sizes(n, len(self._gens)):
            g = self._gens[r]
            with torch.cuda.device(g._model.device):
                z, noise = g.draw_indexed(first_index + lo, hi - lo, seed)
                parts.append(self._generate_on(r, z, noise))
        return self._collect(parts, n, out)

    def _check_out(self, out, n, dev):
        R, nc = 2 ** self.max_res_log2, self.netG.nc
        img, mask = out
        for t, shape in ((img, (n, R, R, nc)), (mask, (n, R, R))):
            if tuple(t.shape) != shape or t.dtype != torch.uint8 or t.device != dev or not t.is_contiguous():
                raise ValueError("out tensors must be contiguous uint8 %s on %s" % (shape, dev))
        return img, mask

    def _generate_on(self, r, z, noise, out=None):
        """The fused step on replica ``r`` (its own device and stream)."""
        g = self._gens[r]
        z, noise, n = g._prepare(z, noise)
        model = g._model
        dev = model.device
        R = 2 ** self.max_res_log2
        if out is None:
            img = torch.empty((n, R, R, g.nc), device=dev, dtype=torch.uint8)
            mask = torch.empty((n, R, R), device=dev, dtype=torch.uint8)
        else:
            img, mask = self._check_out(out, n, dev)
        nptrs = [a.data_ptr() for a in noise]
        if self._graph_wanted(model, n):
            # A small step is a latency chain of ~100 launches of 10-60 us: when the very same call comes again (same batch,
            # same input and output addresses -- a steady loop over preallocated or recycled tensors) often enough it is replayed
            # from a captured hipGraph (+2-3 % at batch <= 2 and in bf16 mode; nothing at batch 8, where it stays eager).  The key holds
            # every pointer the graph bakes in, so a replay always reads the current inputs and writes the current outputs; the
            # epoch changes whenever the context's workspace, weights or stream structure change AND whenever a call of the
            # context failed (a failed pass leaves statistic rows the next EAGER pass re-zeroes -- a replay would not).
            key = (n, z.data_ptr(), tuple(nptrs), img.data_ptr(), mask.data_ptr(), torch.cuda.current_stream(dev).cuda_stream,
                   model.ctx.graph_epoch)
            cache = model.__dict__.setdefault("_graphs", {})
            hit = cache.get(key)
            if hit is not None:
                hit.replay()
                return img, mask
            seen = model.__dict__.setdefault("_graph_seen", {})
            seen[key] = seen.get(key, 0) + 1
            # Capturing costs about as much as a few steps: a call is captured only after it has come `graph_after` (32) times
            # -- a steady loop (main.py generate: thousands of steps), never a one-off call -- and a model captures at most
            # `_GRAPH_CAPTURES_MAX` times in its life (a caller whose addresses keep changing stays eager instead of re-capturing)
            if (seen[key] >= self.graph_after and model.ctx._checked_first_step
                    and model.__dict__.get("_graph_captures", 0) < _GRAPH_CAPTURES_MAX):
                if len(cache) >= 8:
                    cache.pop(next(iter(cache)))
                if len(seen) > 64:
                    seen.clear()
                graph = self._capture(model, dev, n, z, nptrs, img, mask)
                model.__dict__["_graph_captures"] = model.__dict__.get("_graph_captures", 0) + 1      # only a capture that succeeded counts
                graph.replay()
                cache[key] = graph
                return img, mask
        model.ctx.generate(current_stream_ptr(dev), n, z.data_ptr(), nptrs, img.data_ptr(), mask.data_ptr())
        return img, mask

    @staticmethod
    def _capture(model, dev, n, z, nptrs, img, mask):
        """Capture one fused step into a hipGraph on a capture stream of our own: ``CUDAGraph.capture_begin/capture_end``
        directly -- not the ``torch.cuda.graph`` context manager, whose device-wide synchronize, ``gc.collect`` and
        ``empty_cache()`` would stall the caller's steady loop and could move its recycled tensors to new addresses."""
        cur = torch.cuda.current_stream(dev)
        side = model.__dict__.get("_capture_stream")
        if side is None:
            side = model.__dict__["_capture_stream"] = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            graph.capture_begin(capture_error_mode="thread_local")
            try:
                model.ctx.generate(side.cuda_stream, n, z.data_ptr(), nptrs, img.data_ptr(), mask.data_ptr())
            except BaseException:
                # the step failed while it was being recorded: end the (now invalid) capture, but let the ORIGINAL error through --
                # capture_end raises on an invalidated capture and would hide it
                try:
                    graph.capture_end()
                except Exception:
                    pass
                cur.wait_stream(side)
                raise
            graph.capture_end()
        cur.wait_stream(side)
        return graph

    def _graph_wanted(self, model, n):
        """hipGraph replay of the fused step: mode "0" never, "1" always, default = where it measured faster (bf16 mode, fp32
        batches of at most 2); never while per-launch profiling events are on.  ``self.graph_mode`` / ``self.graph_after``
        override the environment's GSA_GRAPH / GSA_GRAPH_AFTER."""
        mode = self.graph_mode if self.graph_mode is not None else os.environ.get("GSA_GRAPH", "")
        if mode == "0" or model.ctx.profiling or len(self._gens) != 1:
            return False
        return mode == "1" or self.precision == "bf16" or n <= 2

    graph_mode = None       # None: GSA_GRAPH decides; "0" / "1"

    @property
    def graph_after(self):
        v = self.__dict__.get("_graph_after")
        return v if v is not None else int(os.environ.get("GSA_GRAPH_AFTER", "32"))

    @graph_after.setter
    def graph_after(self, v):
        self.__dict__["_graph_after"] = v

    def graphs_captured(self):
        """Number of hipGraphs the replicas hold (bench.py reports whether its timed loop replayed one)."""
        return sum(len(g._model.__dict__.get("_graphs", {})) for g in self._gens)

    def snapshot_status(self):
        """In-flight form of the device-side checks (include/gsa.h gsa_status_snapshot): enqueues, behind the kernels of the
        batch just submitted on every replica's current stream, an 8-byte copy of the context's two sticky words into a pinned
        host slot and returns those slots ([2] int32 tensors, one per replica) WITHOUT synchronising.  The reader -- the
        ``DatasetWriter`` dispatcher, before it releases that batch's files -- looks at them once an event recorded behind this
        call has completed; non-zero = that batch (or an earlier one since the last clean look) must be discarded."""
        ring = self.__dict__.get("_status_ring")
        if ring is None:
            ring = self.__dict__["_status_ring"] = [torch.zeros((STATUS_RING_DEPTH, 2), dtype=torch.int32).pin_memory() for _ in self._gens]
            self.__dict__["_status_next"] = 0
        k = self.__dict__["_status_next"]
        self.__dict__["_status_next"] = (k + 1) % STATUS_RING_DEPTH
        views = []
        dev0 = self._gens[0]._model.device
        for g, slots in zip(self._gens, ring):
            dev = g._model.device
            view = slots[k]
            with torch.cuda.device(dev):
                g._model.ctx.status_snapshot(current_stream_ptr(dev), view.data_ptr())
                if dev != dev0:
                    # the reader synchronises with the FIRST device's stream (where the collected pairs live): make that stream
                    # wait for this replica's copy too, or the slot could be read before it has landed
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    torch.cuda.current_stream(dev0).wait_event(ev)
            views.append(view)
        return views

    def _collect(self, parts, n, out):
        """Per-device results -> one (img, mask) pair on the first device, in sample order."""
        dev0 = self._gens[0]._model.device
        img = torch.cat([p[0].to(dev0, non_blocking=True) for p in parts], dim=0)
        mask = torch.cat([p[1].to(dev0, non_blocking=True) for p in parts], dim=0)
        if out is not None:
            oi, om = self._check_out(out, n, dev0)
            oi.copy_(img)
            om.copy_(mask)
            return oi, om
        return img, mask

    def generate_batch(self, z, noise=None, out=None):
        """latents (N,512) [+ noise planes] -> (img (N,R,R,3) u8, mask (N,R,R) u8) on the GPU.
        The per-batch body of ``main.py generate`` (reference main.py:97-99) in one call.
        ``out=(img, mask)``: write into these contiguous uint8 device tensors instead of new ones
        (e.g. the fused send buffer of ``dist.PairGatherer``).  With several gpu ids the batch is split over the
        replicas like the reference's ``split_and_load`` and the pairs are collected on the first device."""
        if self._decoder is None:
            raise RuntimeError("attach_decoder() first")
        if len(self._gens) == 1:
            return self._generate_on(0, z, noise, out)
        n = len(z)
        parts = []
        for r, lo, hi in split_sizes(n, len(self._gens)):
            with torch.cuda.device(self._gens[r]._model.device):
                parts.append(self._generate_on(r, z[lo:hi], None if noise is None else [a[lo:hi] for a in noise]))
        return self._collect(parts, n, out)
