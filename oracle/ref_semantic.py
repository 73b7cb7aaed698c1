"""TEST INFRASTRUCTURE ONLY -- semantic CPU oracle (torch-CPU functionals, NCHW).

Op-for-op restatement of the reference's `generate` hot path in the layer order and
tensor layout the reference uses.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; the product path
(gan-segmentation_amd/) never does and fails loudly without its HIP library.

PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden vectors or
fixtures, its arithmetic lives in Apache MXNet 1.5.1
(reference README.md:9), which is not installable here, and no pretrained weights
are in the tree.  This restatement follows the operator semantics of SURVEY.md
Appendix A and is cross-checked against an independent explicit-loop C
restatement (oracle/c/gsa_oracle.c); golden vectors under tests/golden/ are its
own outputs (tests/golden/make_golden.py).

Follows:
  reference networks_stylegan.py:6-73    StyleGeneratorBlock
  reference networks_stylegan.py:76-197  Generator (mapping, truncation lerp, blocks, toRGB)
  reference networks_stylegan.py:200-236 Blur, :239-264 AdaIN, :267-305 AddNoise,
            :308-315 UpSample, :354-476 Conv2DW / Conv2DTransposeW, :479-545 DenseW / Bias,
            :558-565 PixelNorm
  reference networks_seg.py:7-113        DecoderResBlock, Decoder
  reference image_generator.py:76-84     _transform_gan_back
  reference seg_solver.py:307-329        SegSolver.predict (argmax)
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


def _nf(cfg, r):
    fmaps = int(cfg["fmap_base"] / (2.0 ** ((r - 1) * cfg["fmap_decay"])))
    return min(fmaps, cfg["fmap_max"])


class SemanticGenerator:
    """``Generator(config)(z, noise) -> (rgb, features)`` (reference networks_stylegan.py:76-197).

    ``noise`` is the explicit list of 2*(max_res_log2-1) planes (B,1,R,R) that the
    reference draws inside AddNoise (:297-300)."""

    def __init__(self, cfg, params, dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.p = {k: _t(v, torch.float32) for k, v in params.items()}

    # -- weight scaling, reference networks_stylegan.py:407-412,513-518 (A.5):
    # W_eff = (W * std) * lr_mult, two fp32 roundings in that order; b_eff = b * lr_mult
    def _w(self, prefix, lr_mult=1.0):
        w = self.p[prefix + "_weight"]
        if self.cfg["use_wscale"]:
            w = w * self.p[prefix + "_std"]
        w = w * np.float32(lr_mult)
        return w.to(self.dtype)

    def _b(self, prefix, lr_mult=1.0):
        return (self.p[prefix + "_bias"] * np.float32(lr_mult)).to(self.dtype)

    def mapping(self, z):
        # PixelNorm (reference :558-565), then 8 x (DenseW lr_mult 0.01 -> LeakyReLU 0.2) (:128-139)
        x = z * torch.rsqrt(torch.mean(z * z, dim=1, keepdim=True) + 1e-8)
        for i in range(8):
            x = F.linear(x, self._w("mp_dense_%d" % i, 0.01), self._b("mp_dense_%d" % i, 0.01))
            x = F.leaky_relu(x, 0.2)
        return x

    def lerp(self, psi, w):
        # reference :158-163: latent_avg*(1-psi) + w*psi
        avg = self.p["latent_avg"].to(self.dtype).reshape(1, -1)
        psi = psi.to(self.dtype)
        return avg * (1 - psi) + w * psi

    def adain(self, x, w, prefix):
        # reference :250-264; InstanceNorm eps 1e-5, biased variance (A.7)
        C = x.shape[1]
        s = F.linear(w, self._w(prefix + "_dense_affine"), self._b(prefix + "_dense_affine"))
        s = s.reshape(-1, 2, C)
        ys, yb = s[:, 0, :, None, None], s[:, 1, :, None, None]
        xn = F.instance_norm(x, weight=self.p[prefix + "_norm_gamma"].to(self.dtype),
                             bias=self.p[prefix + "_norm_beta"].to(self.dtype), eps=1e-5)
        return xn * (ys + 1) + yb

    def noise_bias_act(self, x, noise, R, k):
        # AddNoise (:302-304) -> Bias (:544) -> LeakyReLU(0.2) (:40,51)
        sf = self.p["%d_noise_%d_scale_factors" % (R, k)].to(self.dtype)
        b = self.p["%d_bias_%d_bias" % (R, k)].to(self.dtype)
        return F.leaky_relu(x + sf * noise + b, 0.2)

    def block(self, r, y, w1, w2, n1, n2):
        # reference :56-73
        R = 2 ** r
        C = _nf(self.cfg, r)
        if r > 2:
            if r >= 7:  # fused upscale = Deconvolution k4 s2 p1 (:14-18,154)
                y = F.conv_transpose2d(y, self._w("%d_deconv_1" % R), stride=2, padding=1)
            else:       # UpSampling(nearest,2) then 3x3 conv (:22-27)
                y = F.interpolate(y, scale_factor=2, mode="nearest")
                y = F.conv2d(y, self._w("%d_conv_1" % R), padding=1)
            y = F.conv2d(y, self.p["%d_blur_1_w_kernel" % R].to(self.dtype), padding=1, groups=C)
        y = self.noise_bias_act(y, n1, R, 1)
        y = self.adain(y, w1, "%d_adain_1" % R)
        y = F.conv2d(y, self._w("%d_conv_2" % R), padding=1)
        y = self.noise_bias_act(y, n2, R, 2)
        y = self.adain(y, w2, "%d_adain_2" % R)
        return y

    def __call__(self, z, noise):
        cfg = self.cfg
        z = _t(z, self.dtype)
        noise = [_t(n, self.dtype) for n in noise]
        B = z.shape[0]
        w = self.mapping(z)
        psi = self.p["truncation_psi"]
        y = self.p["constant_tensor"].to(self.dtype).expand(B, -1, -1, -1)
        feats = []
        for r in range(2, cfg["max_res_log2"] + 1):
            l = 2 * (r - 2)
            w1, w2 = self.lerp(psi[l], w), self.lerp(psi[l + 1], w)
            y = self.block(r, y, w1, w2, noise[l], noise[l + 1])
            feats.append(y)
        R = 2 ** cfg["max_res_log2"]
        rgb = F.conv2d(y, self._w("%d_conv_to_rgb" % R), self._b("%d_conv_to_rgb" % R))
        return rgb, feats


def transform_gan_back(rgb, imrange=(-1, 1)):
    """reference image_generator.py:76-84: NCHW -> NHWC, to [0,1], clip, x255, truncate to u8."""
    img = np.transpose(np.asarray(rgb, dtype=np.float32), (0, 2, 3, 1))
    img = (img - np.float32(imrange[0])) / np.float32(imrange[1] - imrange[0])
    img = np.clip(img, 0.0, 1.0)
    img = np.float32(255.0) * img
    return img.astype(np.uint8)


class SemanticDecoder:
    """``Decoder(cfg)(*features) -> logits`` at inference (reference networks_seg.py:49-113)."""

    def __init__(self, cfg, params, dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.p = {k: _t(v, dtype) for k, v in params.items()}

    def conv_bn_act(self, x, conv, bn, pad=1):
        p = self.p
        y = F.conv2d(x, p[conv + ".weight"], p[conv + ".bias"], padding=pad)
        if bn is not None:  # nn.BatchNorm at inference, eps 1e-5 (A.13)
            y = F.batch_norm(y, p[bn + ".running_mean"], p[bn + ".running_var"],
                             p[bn + ".gamma"], p[bn + ".beta"], training=False, eps=1e-5)
        return F.leaky_relu(y, 0.2)

    def __call__(self, *features):
        cfg = self.cfg
        Fs, I = cfg["features"], cfg["in_channels"]
        n, s0, use_bn = len(I), cfg["start_res"], cfg["use_bn"]
        prev = None
        for i in range(s0, n):
            x = _t(features[i], self.dtype)
            # cvt_block: conv3x3+bias -> BN -> LeakyReLU -> Dropout(identity) (reference :64-79)
            x = self.conv_bn_act(x, "cvt_block_%d.0" % i, "cvt_block_%d.1" % i if use_bn else None)
            if i > s0:
                x = torch.cat([prev, x], dim=1)  # reference :108-109
            if i < n - 1:
                # main_block: nearest x2 -> DecoderResBlock (reference :7-46,86-88)
                x = F.interpolate(x, scale_factor=2, mode="nearest")
                b = "main_block_%d.1.base_layers" % i
                second = 3 if use_bn else 2
                y = self.conv_bn_act(x, b + ".0", b + ".1" if use_bn else None)
                y = self.conv_bn_act(y, b + ".%d" % second, b + ".%d" % (second + 1) if use_bn else None)
                sc = "main_block_%d.1.shortcut.0" % i
                if sc + ".weight" in self.p:
                    x = F.conv2d(x, self.p[sc + ".weight"], self.p[sc + ".bias"])
                prev = x + y
            else:
                prev = F.conv2d(x, self.p["main_block_%d.0.weight" % i],
                                self.p["main_block_%d.0.bias" % i], padding=1)
        return prev


def predict_mask(logits):
    """reference seg_solver.py:326-327: argmax over classes (first max wins), (N,H,W,1) float32."""
    m = torch.argmax(logits, dim=1, keepdim=True)
    return m.permute(0, 2, 3, 1).to(torch.float32).numpy()


def generate(gcfg, gparams, dcfg, dparams, z, noise, dtype=torch.float32):
    """One batch of the hot path: (img u8 NHWC, mask u8 NHW, rgb f32 NCHW, feats, logits)."""
    with torch.no_grad():
        rgb, feats = SemanticGenerator(gcfg, gparams, dtype)(z, noise)
        logits = SemanticDecoder(dcfg, dparams, dtype)(*feats)
    img = transform_gan_back(rgb.to(torch.float32).numpy())
    mask = predict_mask(logits)[..., 0].astype(np.uint8)
    return img, mask, rgb.numpy(), [f.numpy() for f in feats], logits.numpy()
