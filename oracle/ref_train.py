"""torch-autograd restatement of one decoder training step (test infrastructure only).

The decoder of reference networks_seg.py:49-113 in training mode (BatchNorm batch statistics, explicit Dropout keep
masks so that the device path can be compared sample for sample), the weighted SoftmaxCELoss of
seg_solver.py:395-407, ``backward()`` and MXNet's Adam update.  CPU, float32."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _bn_lrelu(x, p, prefix, running, momentum=0.9):
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    running[prefix + ".running_mean"] = momentum * p[prefix + ".running_mean"] + (1 - momentum) * mean.detach()
    running[prefix + ".running_var"] = momentum * p[prefix + ".running_var"] + (1 - momentum) * var.detach()
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5)
    return F.leaky_relu(xh * p[prefix + ".gamma"][None, :, None, None] + p[prefix + ".beta"][None, :, None, None], 0.2)


def forward_train(cfg, p, feats, masks, keep=0.5):
    """-> (logits, {new running statistics})."""
    nl, s0 = len(cfg["in_channels"]), cfg["start_res"]
    running = {}
    prev = None
    for i in range(s0, nl):
        cv = "cvt_block_%d" % i
        x = F.conv2d(feats[i], p[cv + ".0.weight"], p[cv + ".0.bias"], padding=1)
        x = _bn_lrelu(x, p, cv + ".1", running)
        if masks[i] is not None:
            x = x * masks[i].float() / keep
        inp = torch.cat([prev, x], 1) if i > s0 else x
        if i < nl - 1:
            b = "main_block_%d.1.base_layers" % i
            up = F.interpolate(inp, scale_factor=2, mode="nearest")
            y = F.conv2d(up, p[b + ".0.weight"], p[b + ".0.bias"], padding=1)
            y = _bn_lrelu(y, p, b + ".1", running)
            y = F.conv2d(y, p[b + ".3.weight"], p[b + ".3.bias"], padding=1)
            y = _bn_lrelu(y, p, b + ".4", running)
            sc = "main_block_%d.1.shortcut.0" % i
            s = F.conv2d(up, p[sc + ".weight"], p[sc + ".bias"]) if sc + ".weight" in p else up
            prev = s + y
        else:
            fn = "main_block_%d.0" % i
            return F.conv2d(inp, p[fn + ".weight"], p[fn + ".bias"], padding=1), running


def train_step(cfg, params, feats, labels, masks, t, m, v, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, keep=0.5):
    """One step on copies: -> (loss per sample, new params, new m, new v).  ``t`` = step count after this step."""
    p = {k: torch.tensor(np.asarray(a, np.float32), requires_grad=not k.endswith(("running_mean", "running_var")))
         for k, a in params.items()}
    feats = [torch.tensor(np.asarray(f, np.float32)) for f in feats]
    labels = torch.tensor(np.asarray(labels)).long()
    n = feats[0].shape[0]
    logits, running = forward_train(cfg, p, feats, [None if mk is None else torch.tensor(np.asarray(mk)) for mk in masks], keep)
    ce = F.cross_entropy(logits, labels.clamp(min=0), reduction="none") * (labels > -1).float()
    per_sample = ce.mean(dim=(1, 2))
    per_sample.sum().backward()
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    new_p, new_m, new_v = {}, {}, {}
    for k, w in p.items():
        if not w.requires_grad:
            new_p[k] = running[k].numpy()
            continue
        g = (w.grad if w.grad is not None else torch.zeros_like(w)) / n
        mm = b1 * torch.tensor(m[k]) + (1 - b1) * g
        vv = b2 * torch.tensor(v[k]) + (1 - b2) * g * g
        new_m[k], new_v[k] = mm.numpy(), vv.numpy()
        new_p[k] = (w.detach() - lr_t * mm / (vv.sqrt() + eps)).numpy()
    return per_sample.detach().numpy(), new_p, new_m, new_v, {k: (w.grad.numpy() if w.grad is not None else None) for k, w in p.items() if w.requires_grad}
