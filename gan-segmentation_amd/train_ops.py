"""ctypes binding of the decoder-training operators (include/gsa_train.h, csrc/gsa_train.hip).

Tensors are torch CUDA tensors used as device memory only (fp32 NCHW, contiguous); every function enqueues HIP
kernels on the current stream of the tensors' device and returns its output tensors.  No CPU fallback."""
import ctypes

import torch

from . import _lib
from ._runtime import current_stream_ptr

_FUNCS = None


def _api():
    global _FUNCS
    if _FUNCS is None:
        lib = _lib.load_library().lib
        c = ctypes
        vp, i32, i64, f32 = c.c_void_p, c.c_int32, c.c_int64, c.c_float
        sig = {
            "gsa_train_conv": [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, vp, i32, vp, i32],
            "gsa_train_conv_wgrad": [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp],
            "gsa_train_bn_lrelu_fwd": [vp, i32, i32, i32, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, f32, vp],
            "gsa_train_bn_lrelu_bwd": [vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, f32, vp, vp, vp],
            "gsa_train_bn_sums": [vp, i32, i32, i32, vp, vp],
            "gsa_train_bn_lrelu_fwd_sums": [vp, i32, i32, i32, c.c_double, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, f32, vp],
            "gsa_train_bn_bwd_sums": [vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, f32, vp, vp],
            "gsa_train_bn_lrelu_bwd_sums": [vp, i32, i32, i32, c.c_double, vp, vp, vp, f32, vp, vp, vp, f32, vp, vp, vp, vp, vp],
            "gsa_train_softmax_ce": [vp, i32, i32, i32, vp, vp, vp, vp, f32],
            "gsa_train_upsample2_bwd": [vp, i32, i32, i32, i32, vp, vp, i32],
            "gsa_train_add": [vp, i64, vp, vp, vp],
            "gsa_train_dropout_mask": [vp, i64, c.c_uint64, c.c_uint32, f32, vp],
            "gsa_train_adam": [vp, i64, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32],
        }
        _FUNCS = {}
        for name, args in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = c.c_int, args
            _FUNCS[name] = fn
    return _FUNCS


def _call(name, *args):
    rc = _api()[name](*args)
    if rc != 0:
        raise _lib.GsaError("%s failed (%d)" % (name, rc))


def _p(t):
    return t.data_ptr() if t is not None else None


def _chk(*tensors):
    for t in tensors:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise ValueError("training ops take contiguous CUDA tensors")


def conv(x0, x1, w, bias=None, up=0, transposed=False, cout0=None, out0=None, out1=None, accumulate=False):
    """Convolution / input gradient over concat(x0, x1) (x1 may be None); see gsa_train_conv.  -> (out0, out1)."""
    _chk(x0, x1, w, bias, out0, out1)
    n, C0, Hs, Ws = x0.shape
    C1 = x1.shape[1] if x1 is not None else 0
    K = w.shape[2]
    Cout = w.shape[1] if transposed else w.shape[0]
    assert (w.shape[0] if transposed else w.shape[1]) == C0 + C1, "weight does not match the input channels"
    cout0 = Cout if cout0 is None else cout0
    H, W = Hs << up, Ws << up
    if out0 is None:
        out0 = torch.empty((n, cout0, H, W), device=x0.device, dtype=torch.float32)
    if cout0 < Cout and out1 is None:
        out1 = torch.empty((n, Cout - cout0, H, W), device=x0.device, dtype=torch.float32)
    _call("gsa_train_conv", current_stream_ptr(x0.device), n, _p(x0), C0, _p(x1), C1, Hs, Ws, int(up), _p(w), Cout, K,
          1 if transposed else 0, _p(bias), _p(out0), cout0, _p(out1), 1 if accumulate else 0)
    return out0, out1


def conv_wgrad(x0, x1, dy, K, dw, db=None, up=0):
    """dw (Cout, C0+C1, K, K) += dL/dW, db (Cout) += dL/db."""
    _chk(x0, x1, dy, dw, db)
    n, C0, Hs, Ws = x0.shape
    C1 = x1.shape[1] if x1 is not None else 0
    _call("gsa_train_conv_wgrad", current_stream_ptr(x0.device), n, _p(x0), C0, _p(x1), C1, Hs, Ws, int(up), _p(dy), dy.shape[1], K,
          _p(dw), _p(db))


def bn_lrelu_fwd(v, gamma, beta, running_mean, running_var, mask=None, drop_scale=1.0, eps=1e-5, momentum=0.9):
    """-> (y, batch mean, batch var); running statistics are updated in place."""
    _chk(v, gamma, beta, running_mean, running_var, mask)
    n, C, H, W = v.shape
    mean = torch.empty(C, device=v.device, dtype=torch.float32)
    var = torch.empty(C, device=v.device, dtype=torch.float32)
    y = torch.empty_like(v)
    _call("gsa_train_bn_lrelu_fwd", current_stream_ptr(v.device), n, C, H * W, _p(v), _p(gamma), _p(beta), eps, momentum, _p(mean), _p(var),
          _p(running_mean), _p(running_var), _p(mask), drop_scale, _p(y))
    return y, mean, var


def bn_lrelu_bwd(v, gamma, beta, mean, var, g, dgamma, dbeta, mask=None, drop_scale=1.0, eps=1e-5):
    """g: dL/dy on entry, dL/dv on exit (in place); dgamma / dbeta accumulated."""
    _chk(v, gamma, beta, mean, var, g, dgamma, dbeta, mask)
    n, C, H, W = v.shape
    _call("gsa_train_bn_lrelu_bwd", current_stream_ptr(v.device), n, C, H * W, _p(v), _p(gamma), _p(beta), eps, _p(mean), _p(var), _p(mask),
          drop_scale, _p(g), _p(dgamma), _p(dbeta))
    return g


def sync_bn_lrelu_fwd(v, gamma, beta, running_mean, running_var, all_reduce, mask=None, drop_scale=1.0, eps=1e-5, momentum=0.9):
    """SyncBatchNorm form of ``bn_lrelu_fwd`` (reference networks_seg.py:20-21: ``gluon.contrib.nn.SyncBatchNorm``): the
    per-channel sums and the pixel count go through ``all_reduce(tensor)`` (in place, sum over the ranks) before the
    statistics are formed.  -> (y, mean, var) with the statistics of the whole (all-rank) batch."""
    _chk(v, gamma, beta, running_mean, running_var, mask)
    n, C, H, W = v.shape
    sums = torch.empty(2 * C + 1, device=v.device, dtype=torch.float64)
    _call("gsa_train_bn_sums", current_stream_ptr(v.device), n, C, H * W, _p(v), _p(sums))
    sums[2 * C] = float(n * H * W)
    all_reduce(sums)
    count = float(sums[2 * C].item())
    mean = torch.empty(C, device=v.device, dtype=torch.float32)
    var = torch.empty(C, device=v.device, dtype=torch.float32)
    y = torch.empty_like(v)
    _call("gsa_train_bn_lrelu_fwd_sums", current_stream_ptr(v.device), n, C, H * W, count, _p(v), _p(gamma), _p(beta), eps, momentum,
          _p(sums), _p(mean), _p(var), _p(running_mean), _p(running_var), _p(mask), drop_scale, _p(y))
    return y, mean, var, count


def sync_bn_lrelu_bwd(v, gamma, beta, mean, var, count, g, dgamma, dbeta, all_reduce, mask=None, drop_scale=1.0, eps=1e-5):
    """Backward of ``sync_bn_lrelu_fwd``: the two batch means of the gradient are taken over all ranks; dgamma / dbeta
    receive this rank's share (the gradient all-reduce adds the ranks)."""
    _chk(v, gamma, beta, mean, var, g, dgamma, dbeta, mask)
    n, C, H, W = v.shape
    own = torch.empty(2 * C, device=v.device, dtype=torch.float64)
    _call("gsa_train_bn_bwd_sums", current_stream_ptr(v.device), n, C, H * W, _p(v), _p(gamma), _p(beta), eps, _p(mean), _p(var), _p(mask),
          drop_scale, _p(g), _p(own))
    both = own.clone()
    all_reduce(both)
    _call("gsa_train_bn_lrelu_bwd_sums", current_stream_ptr(v.device), n, C, H * W, float(count), _p(v), _p(gamma), _p(beta), eps, _p(mean),
          _p(var), _p(mask), drop_scale, _p(both), _p(own), _p(g), _p(dgamma), _p(dbeta))
    return g


def softmax_ce(logits, labels, grad_scale=1.0):
    """-> (per-sample loss (n,), dlogits)."""
    _chk(logits, labels)
    n, K, H, W = logits.shape
    loss = torch.empty(n, device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits)
    _call("gsa_train_softmax_ce", current_stream_ptr(logits.device), n, K, H * W, _p(logits), _p(labels), _p(loss), _p(dlogits), grad_scale)
    return loss, dlogits


def upsample2_bwd(dy_up, dx=None, accumulate=False):
    _chk(dy_up, dx)
    n, C, H2, W2 = dy_up.shape
    if dx is None:
        dx = torch.empty((n, C, H2 // 2, W2 // 2), device=dy_up.device, dtype=torch.float32)
    _call("gsa_train_upsample2_bwd", current_stream_ptr(dy_up.device), n, C, H2 // 2, W2 // 2, _p(dy_up), _p(dx), 1 if accumulate else 0)
    return dx


def add(a, b, out=None):
    _chk(a, b, out)
    out = torch.empty_like(a) if out is None else out
    _call("gsa_train_add", current_stream_ptr(a.device), a.numel(), _p(a), _p(b), _p(out))
    return out


def dropout_mask(shape, seed, stream_id, keep_prob, device):
    mask = torch.empty(shape, device=device, dtype=torch.uint8)
    _call("gsa_train_dropout_mask", current_stream_ptr(mask.device), mask.numel(), int(seed) & (2 ** 64 - 1), int(stream_id) & 0xFFFFFFFF,
          keep_prob, _p(mask))
    return mask


def adam(w, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, rescale=1.0, wd=0.0):
    _chk(w, g, m, v)
    _call("gsa_train_adam", current_stream_ptr(w.device), w.numel(), _p(w), _p(g), _p(m), _p(v), lr_t, beta1, beta2, eps, rescale, wd)
