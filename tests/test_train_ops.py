"""Decoder-training operators (include/gsa_train.h) against torch functionals / autograd on the CPU
(SURVEY.md section 8f-3).  Tolerances: fp32 sums in a different order (float atomics) -> 1e-4 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def _close(a, b, tol=2e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(1.0, float(b.abs().max()))
    assert float((a - b).abs().max()) <= tol * scale, "max |d| %.3e vs scale %.3e" % (float((a - b).abs().max()), scale)


@pytest.mark.parametrize("C0,C1,Cout,K,up,R", [(16, 0, 16, 3, 0, 20), (8, 8, 12, 3, 1, 8), (32, 32, 2, 3, 0, 16), (24, 8, 16, 1, 1, 8),
                                               (512, 0, 32, 3, 0, 4), (16, 16, 16, 3, 1, 40), (16, 0, 32, 3, 0, 80),
                                               (40, 0, 24, 1, 0, 48),
                                               (8, 4, 6, 3, 0, 6), (6, 0, 4, 1, 1, 5)])     # W % 4 != 0: the vector-ALU fallbacks
def test_conv_forward_dgrad_wgrad(T, C0, C1, Cout, K, up, R):
    import torch.nn.functional as F
    from gan_segmentation_amd import train_ops as ops
    g = T.Generator().manual_seed(C0 + Cout + R)
    n = 2
    x0 = T.randn(n, C0, R, R, generator=g)
    x1 = T.randn(n, C1, R, R, generator=g) if C1 else None
    w = T.randn(Cout, C0 + C1, K, K, generator=g) * 0.1
    b = T.randn(Cout, generator=g)
    xin = T.cat([x0, x1], 1) if C1 else x0
    xin = xin.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    xu = F.interpolate(xin, scale_factor=2, mode="nearest") if up else xin
    y = F.conv2d(xu, wr, br, padding=K // 2)
    dy = T.randn(y.shape, generator=g)
    y.backward(dy)
    dev = "cuda"
    d = lambda t: t.to(dev).contiguous() if t is not None else None
    out, _ = ops.conv(d(x0), d(x1), d(w), d(b), up=up)
    _close(out, y)
    # input gradient w.r.t. the (upsampled) input, split back over the two sources
    dxu0, dxu1 = ops.conv(d(dy), None, d(w), None, transposed=True, cout0=C0)
    dx0 = ops.upsample2_bwd(dxu0) if up else dxu0
    _close(dx0, xin.grad[:, :C0])
    if C1:
        dx1 = ops.upsample2_bwd(dxu1) if up else dxu1
        _close(dx1, xin.grad[:, C0:])
    dw = T.zeros_like(w).to(dev)
    db = T.zeros_like(b).to(dev)
    ops.conv_wgrad(d(x0), d(x1), d(dy), K, dw, db, up=up)
    _close(dw, wr.grad, 5e-4)
    _close(db, br.grad, 5e-4)
    # accumulate flag
    out2, _ = ops.conv(d(x0), d(x1), d(w), d(b), up=up, out0=out.clone(), accumulate=True)
    _close(out2, 2 * y)


def test_bn_lrelu_dropout_forward_backward(T):
    import torch.nn.functional as F
    from gan_segmentation_amd import train_ops as ops
    g = T.Generator().manual_seed(3)
    n, C, R = 2, 12, 24
    v = T.randn(n, C, R, R, generator=g) * 2 + 0.5
    gamma = T.rand(C, generator=g) + 0.5
    beta = T.randn(C, generator=g) * 0.1
    rm, rv = T.randn(C, generator=g) * 0.1, T.rand(C, generator=g) + 0.5
    mask = (T.rand(n, C, R, R, generator=g) < 0.5).to(T.uint8)
    vr, gr, br = v.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_t, rv_t = rm.clone(), rv.clone()
    # torch momentum = 1 - mxnet momentum; torch keeps the UNBIASED variance in running_var, MXNet the biased one
    z = F.batch_norm(vr, rm_t, rv_t, gr, br, training=True, momentum=0.1, eps=1e-5)
    y = F.leaky_relu(z, 0.2) * mask.float() * 2.0
    dy = T.randn(y.shape, generator=g)
    y.backward(dy)
    d = lambda t: t.cuda().contiguous()
    rm_d, rv_d = d(rm), d(rv)
    yd, mean, var = ops.bn_lrelu_fwd(d(v), d(gamma), d(beta), rm_d, rv_d, mask=d(mask), drop_scale=2.0)
    _close(yd, y)
    _close(mean, v.mean(dim=(0, 2, 3)))
    _close(var, v.var(dim=(0, 2, 3), unbiased=False))
    _close(rm_d, rm_t)
    _close(rv_d, rv * 0.9 + v.var(dim=(0, 2, 3), unbiased=False) * 0.1)
    gd = d(dy)
    dgam, dbet = T.zeros(C).cuda(), T.zeros(C).cuda()
    ops.bn_lrelu_bwd(d(v), d(gamma), d(beta), mean, var, gd, dgam, dbet, mask=d(mask), drop_scale=2.0)
    _close(gd, vr.grad, 5e-4)
    _close(dgam, gr.grad, 5e-4)
    _close(dbet, br.grad, 5e-4)


def test_softmax_ce_adam_mask_add(T):
    import torch.nn.functional as F
    from gan_segmentation_amd import train_ops as ops
    g = T.Generator().manual_seed(5)
    n, K, R = 3, 2, 32
    logits = T.randn(n, K, R, R, generator=g) * 2
    labels = T.randint(0, K, (n, R, R), generator=g)
    labels[T.rand(n, R, R, generator=g) < 0.25] = -1
    lr = logits.clone().requires_grad_(True)
    ce = F.cross_entropy(lr, labels.clamp(min=0), reduction="none") * (labels > -1).float()
    per_sample = ce.mean(dim=(1, 2))
    per_sample.sum().backward()
    loss, dl = ops.softmax_ce(logits.cuda(), labels.to(T.int8).cuda())
    _close(loss, per_sample)
    _close(dl, lr.grad, 1e-5)
    # Adam (MXNet form)
    w, gr_, m, v = [T.randn(1000, generator=g) for _ in range(4)]
    v = v.abs()
    t, lr0, b1, b2, eps = 3, 1e-3, 0.9, 0.999, 1e-8
    lr_t = lr0 * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    gg = gr_ * 0.5
    m2 = b1 * m + (1 - b1) * gg
    v2 = b2 * v + (1 - b2) * gg * gg
    w2 = w - lr_t * m2 / (v2.sqrt() + eps)
    wd_, md, vd = w.cuda(), m.cuda(), v.cuda()
    ops.adam(wd_, gr_.cuda(), md, vd, float(lr_t), b1, b2, eps, rescale=0.5)
    _close(wd_, w2, 1e-6); _close(md, m2, 1e-6); _close(vd, v2, 1e-6)
    # dropout mask: deterministic in (seed, stream), keep probability respected
    m1 = ops.dropout_mask((4, 8, 64, 64), 11, 7, 0.5, "cuda")
    m2_ = ops.dropout_mask((4, 8, 64, 64), 11, 7, 0.5, "cuda")
    m3 = ops.dropout_mask((4, 8, 64, 64), 11, 8, 0.5, "cuda")
    assert T.equal(m1, m2_) and not T.equal(m1, m3)
    assert abs(float(m1.float().mean()) - 0.5) < 0.01
    a, b = T.randn(1000, generator=g), T.randn(1000, generator=g)
    _close(ops.add(a.cuda(), b.cuda()), a + b, 1e-7)
