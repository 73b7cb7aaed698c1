"""Decoder training on the MI355X: the graph of reference ``SegSolver.fit``
(seg_solver.py:351-465) -- ``self.net(*features)`` in training mode, weighted SoftmaxCELoss, ``backward()``,
``trainer.step`` with Adam -- written out by hand over the HIP operators of ``train_ops`` (include/gsa_train.h).

The generator features are inputs only (the reference ``.detach()``es them, :397), so the backward pass stops
at the per-feature ``cvt`` convolutions.  Training-mode semantics reproduced: BatchNorm uses batch statistics
and updates its running statistics with momentum 0.9 (MXNet ``nn.BatchNorm`` defaults; biased variance),
Dropout(0.5) in every ``cvt`` block (networks_seg.py:76-78) with a counter-based mask keyed on (seed, step,
level), Adam as ``mx.optimizer.Adam`` (bias correction folded into the learning rate, eps outside the root),
gradients rescaled by 1/batch (``trainer.step(batch_size)``).

PyTorch provides the device buffers; every number is computed by a HIP kernel of this project.
"""
import math

import numpy as np
import torch

from . import train_ops as ops
from . import weights as _weights
from ._runtime import require_gpu


class DecoderTrainer:
    def __init__(self, cfg, params, device=0, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, seed=1,
                 dropout_keep=0.5, sync_bn=None, all_reduce=None, world=None):
        require_gpu()
        if not cfg["use_bn"]:
            raise NotImplementedError("training supports the reference's configuration: use_bn=True")
        self.cfg = dict(cfg)
        self.dev = torch.device("cuda", device)
        self.F, self.I = list(cfg["features"]), list(cfg["in_channels"])
        self.n_levels = len(self.I)
        self.s0 = int(cfg["start_res"])     # first feature consumed (reference networks_seg.py:56)
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, wd
        self.seed = seed
        self.keep = dropout_keep if cfg.get("use_dropout", True) else 1.0
        self.t = 0
        self._eyes = {}
        # SyncBatchNorm (reference networks_seg.py:20-21,30-31,73-74 under cfg['use_sync_bn']; off in seg_solver.py:120):
        # batch statistics and their gradient means over the batches of ALL ranks, exchanged as per-channel sums
        self.sync_bn = bool(cfg.get("use_sync_bn", False)) if sync_bn is None else bool(sync_bn)
        self._all_reduce = all_reduce if all_reduce is not None else self._dist_all_reduce
        self._world_override = world    # with a caller-supplied all_reduce: the number of ranks it sums over
        shapes = _weights.decoder_param_shapes(cfg)
        self.p, self.g, self.m, self.v = {}, {}, {}, {}
        for name, shape in shapes.items():
            a = np.ascontiguousarray(params[name], np.float32).reshape(shape)
            self.p[name] = torch.from_numpy(a.copy()).to(self.dev)
            if not name.endswith(("running_mean", "running_var")):
                self.g[name] = torch.zeros(shape, device=self.dev)
                self.m[name] = torch.zeros(shape, device=self.dev)
                self.v[name] = torch.zeros(shape, device=self.dev)

    # -- helpers ----------------------------------------------------------------------------------
    @staticmethod
    def _dist_all_reduce(t):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            torch.distributed.all_reduce(t)

    def _bn_fwd(self, prefix, v, mask=None):
        p = self.p
        if self.sync_bn:
            y, mean, var, count = ops.sync_bn_lrelu_fwd(v, p[prefix + ".gamma"], p[prefix + ".beta"], p[prefix + ".running_mean"],
                                                        p[prefix + ".running_var"], self._all_reduce, mask=mask,
                                                        drop_scale=1.0 / self.keep)
            return y, (mean, var, count)
        y, mean, var = ops.bn_lrelu_fwd(v, p[prefix + ".gamma"], p[prefix + ".beta"], p[prefix + ".running_mean"],
                                        p[prefix + ".running_var"], mask=mask, drop_scale=1.0 / self.keep)
        return y, (mean, var)

    def _bn_bwd(self, prefix, v, stats, g, mask=None):
        p = self.p
        if self.sync_bn:
            return ops.sync_bn_lrelu_bwd(v, p[prefix + ".gamma"], p[prefix + ".beta"], stats[0], stats[1], stats[2], g,
                                         self.g[prefix + ".gamma"], self.g[prefix + ".beta"], self._all_reduce, mask=mask,
                                         drop_scale=1.0 / self.keep)
        return ops.bn_lrelu_bwd(v, p[prefix + ".gamma"], p[prefix + ".beta"], stats[0], stats[1], g, self.g[prefix + ".gamma"],
                                self.g[prefix + ".beta"], mask=mask, drop_scale=1.0 / self.keep)

    def dropout_masks(self, shapes):
        """The keep masks of this step, one per level (None when dropout is off)."""
        if self.keep >= 1.0:
            return [None] * len(shapes)
        return [ops.dropout_mask(s, self.seed, (self.t << 8) | i, self.keep, self.dev) for i, s in enumerate(shapes)]

    # -- one optimisation step --------------------------------------------------------------------
    def step(self, features, labels, masks=None):
        """features: list of (n,C_i,R_i,R_i) fp32 arrays/tensors; labels (n,H,W) integers, -1 = ignore.
        ``masks``: explicit dropout keep masks (tests); default: counter-based.  -> mean loss of the batch."""
        with torch.cuda.device(self.dev):    # the training C ABI is stateless: launches go to the current device
            return self._step(features, labels, masks)

    def _step(self, features, labels, masks):
        p, g, F, nl, s0 = self.p, self.g, self.F, self.n_levels, self.s0
        feats = [torch.as_tensor(np.asarray(f) if not torch.is_tensor(f) else f, dtype=torch.float32).to(self.dev).contiguous()
                 for f in features]
        if feats[0].dim() == 3:
            feats = [f.unsqueeze(0).contiguous() for f in feats]
        n = feats[0].shape[0]
        lab = torch.as_tensor(np.asarray(labels) if not torch.is_tensor(labels) else labels).reshape(n, *feats[-1].shape[2:])
        lab = lab.to(device=self.dev, dtype=torch.int8).contiguous()
        for t in g.values():
            t.zero_()
        if masks is None:
            masks = self.dropout_masks([(n, F[i], f.shape[2], f.shape[3]) for i, f in enumerate(feats)])
        self.t += 1

        # ---- forward (training mode), keeping what the backward pass needs
        saved = {}
        prev = None
        logits = None
        for i in range(s0, nl):
            cv = "cvt_block_%d" % i
            cv_raw, _ = ops.conv(feats[i], None, p[cv + ".0.weight"], p[cv + ".0.bias"])
            cvt, cv_stats = self._bn_fwd(cv + ".1", cv_raw, masks[i])
            src0, src1 = (prev, cvt) if i > s0 else (cvt, None)
            rec = {"cv_raw": cv_raw, "cv_stats": cv_stats, "src0": src0, "src1": src1}
            if i < nl - 1:
                b = "main_block_%d.1.base_layers" % i
                a_raw, _ = ops.conv(src0, src1, p[b + ".0.weight"], p[b + ".0.bias"], up=1)
                a, a_stats = self._bn_fwd(b + ".1", a_raw)
                b_raw, _ = ops.conv(a, None, p[b + ".3.weight"], p[b + ".3.bias"])
                y, b_stats = self._bn_fwd(b + ".4", b_raw)
                sc_name = "main_block_%d.1.shortcut.0" % i
                if sc_name + ".weight" in p:
                    sc, _ = ops.conv(src0, src1, p[sc_name + ".weight"], p[sc_name + ".bias"], up=1)
                    prev = ops.add(sc, y)
                else:   # identity shortcut on the upsampled (possibly concatenated) input: in_c == conv_size
                    prev = self._identity_up_add(src0, src1, y)
                rec.update(a_raw=a_raw, a=a, a_stats=a_stats, b_raw=b_raw, b_stats=b_stats)
            else:
                fn = "main_block_%d.0" % i
                logits, _ = ops.conv(src0, src1, p[fn + ".weight"], p[fn + ".bias"])
            saved[i] = rec

        loss, dlogits = ops.softmax_ce(logits, lab)

        # ---- backward
        dprev = None        # gradient w.r.t. the output of main block i-1 (`prev` of level i)
        for i in reversed(range(s0, nl)):
            rec = saved[i]
            src0, src1 = rec["src0"], rec["src1"]
            C0 = src0.shape[1]
            if i == nl - 1:
                fn = "main_block_%d.0" % i
                ops.conv_wgrad(src0, src1, dlogits, 3, g[fn + ".weight"], g[fn + ".bias"])
                d0, d1 = ops.conv(dlogits, None, p[fn + ".weight"], transposed=True, cout0=C0)
            else:
                b = "main_block_%d.1.base_layers" % i
                dy = dprev                                              # dL/d(sc + y)
                gy = self._bn_bwd(b + ".4", rec["b_raw"], rec["b_stats"], dy.clone())
                ops.conv_wgrad(rec["a"], None, gy, 3, g[b + ".3.weight"], g[b + ".3.bias"])
                ga, _ = ops.conv(gy, None, p[b + ".3.weight"], transposed=True)
                ga = self._bn_bwd(b + ".1", rec["a_raw"], rec["a_stats"], ga)
                ops.conv_wgrad(src0, src1, ga, 3, g[b + ".0.weight"], g[b + ".0.bias"], up=1)
                u0, u1 = ops.conv(ga, None, p[b + ".0.weight"], transposed=True, cout0=C0)     # w.r.t. the upsampled input
                sc_name = "main_block_%d.1.shortcut.0" % i
                if sc_name + ".weight" in p:
                    ops.conv_wgrad(src0, src1, dy, 1, g[sc_name + ".weight"], g[sc_name + ".bias"], up=1)
                    ops.conv(dy, None, p[sc_name + ".weight"], transposed=True, cout0=C0, out0=u0, out1=u1, accumulate=True)
                else:       # identity shortcut: dL/d(upsampled input) += dy, channel by channel over the two sources
                    ops.conv(dy, None, self._eye(dy.shape[1]), transposed=True, cout0=C0, out0=u0, out1=u1, accumulate=True)
                d0 = ops.upsample2_bwd(u0)
                d1 = ops.upsample2_bwd(u1) if u1 is not None else None
            dcvt, dprev = (d1, d0) if i > s0 else (d0, None)
            cv = "cvt_block_%d" % i
            gv = self._bn_bwd(cv + ".1", rec["cv_raw"], rec["cv_stats"], dcvt, masks[i])
            ops.conv_wgrad(feats[i], None, gv, 3, g[cv + ".0.weight"], g[cv + ".0.bias"])

        # ---- data parallel (one process per GPU): sum the gradients over the ranks with RCCL -- the reference's
        # kvstore 'nccl' (seg_solver.py:55); BatchNorm statistics stay per rank unless cfg['use_sync_bn'] (off in the reference, :120)
        world = self._world_override
        if world is None:
            world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        if world > 1:
            for name in g:
                self._all_reduce(g[name])
        # ---- Adam (mx.optimizer.Adam; trainer.step(batch_size) -> rescale_grad = 1/batch)
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for name in g:
            ops.adam(p[name], g[name], self.m[name], self.v[name], lr_t, self.b1, self.b2, self.eps, rescale=1.0 / (n * world), wd=self.wd)
        return float(loss.cpu().numpy().mean())

    def _eye(self, C):
        e = self._eyes.get(C)
        if e is None:
            e = self._eyes[C] = torch.eye(C, device=self.dev).reshape(C, C, 1, 1).contiguous()
        return e

    def _identity_up_add(self, x0, x1, y):
        """y + nearest-x2(concat(x0, x1)) without a dedicated kernel: a 1x1 convolution with the identity matrix on the
        upsampled read (level 0 of the reference's decoder: 32 -> 32 channels at 4x4 -> 8x8; x1 is None there)."""
        out = y.clone()
        ops.conv(x0, x1, self._eye(y.shape[1]), None, up=1, out0=out, accumulate=True)
        return out

    # -- parameters -------------------------------------------------------------------------------
    def state_dict(self):
        """{structural name: numpy array} -- what ``Decoder.load_parameters`` / ``save_params`` take."""
        torch.cuda.synchronize(self.dev)
        return {k: v.detach().cpu().numpy().copy() for k, v in self.p.items()}
