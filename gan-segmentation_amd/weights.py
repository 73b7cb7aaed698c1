"""Network configurations, parameter names/shapes and a synthetic weight factory.

Names and shapes are the weight-layout contract of the reference (SURVEY.md
Appendix B):

* generator, "scheme P" (prefix names, what ``collect_params()`` yields; the
  ``prefix=`` arguments in reference networks_stylegan.py:16-53,123-137,244-247)
* generator, "scheme S" (structural names written by ``save_parameters``)
* decoder, scheme S (reference networks_seg.py:14-92, written by
  seg_solver.py:331-337)

No pretrained weights exist in this environment (reference README.md:18,20 are
external downloads), so parity and benchmarks run on synthetic weights drawn in
the regime the layers were designed for.
"""
import math
import re

import numpy as np

GAN_MAX_RES_LOG2 = {"ffhq": 10, "cars": 9, "bedrooms": 8}  # reference image_generator.py:11


def generator_config(max_res_log2=10, **overrides):
    """The fixed generator config of reference image_generator.py:46-74."""
    cfg = {
        "use_wscale": True, "fmap_base": 8192, "fmap_decay": 1.0, "fmap_max": 512,
        "max_res_log2": max_res_log2, "fix_noise": False,
        "base_scale_x": 4, "base_scale_y": 4,
        "init": "normal", "init_normal_std": 1.0, "init_xavier_magnitude": 1.0,
        "latent_size": 512, "latent_prior": "normal",
        "channels": 3, "imrange": (-1, 1), "dtype": "fp32",
    }
    cfg.update(overrides)
    return cfg


def reduced_generator_config(max_res_log2=7):
    """A small config the reference's own knobs allow (fmap_base/fmap_max):
    channels 64,64,64,64,32,16 at 4..128 px.  Covers both conv_1 variants
    (nearest-up + conv3x3 below 128 px, Deconvolution 4x4 s2 from 128 px up,
    reference networks_stylegan.py:154)."""
    return generator_config(max_res_log2=max_res_log2, fmap_base=1024, fmap_max=64)


def num_features(cfg, res_log2):
    # reference networks_stylegan.py:114-116
    fmaps = int(cfg["fmap_base"] / (2.0 ** ((res_log2 - 1) * cfg["fmap_decay"])))
    return min(fmaps, cfg["fmap_max"])


def generator_channels(cfg):
    return [num_features(cfg, r) for r in range(2, cfg["max_res_log2"] + 1)]


def num_style_layers(cfg):
    return 2 * (cfg["max_res_log2"] - 1)


def decoder_config(max_res_log2=10, num_classes=2, in_channels=None):
    """Inference-relevant part of reference seg_solver.py:83-132."""
    features = [32, 32, 32, 32, 32, 32, 32, 32, 16][:max_res_log2 - 1] + [num_classes]
    if in_channels is None:
        in_channels = [512, 512, 512, 512, 256, 128, 64, 32, 16][:max_res_log2 - 1]
    return {
        "num_classes": num_classes, "use_bn": True, "use_sync_bn": False,
        "use_dropout": True, "start_res": 0,
        "features": features, "in_channels": list(in_channels), "dtype": "fp32",
    }


def _std(gain, fan_in):
    # float64 formula cast to fp32 by get_constant (reference networks_stylegan.py:399-403,506-509)
    return np.array([gain / np.sqrt(fan_in)], dtype=np.float32)


def generator_param_shapes(cfg):
    """Ordered ``{scheme-P name: shape}`` of every parameter ``Generator`` declares."""
    L = num_style_layers(cfg)
    ls = cfg["latent_size"]
    c2 = num_features(cfg, 2)
    shapes = {
        "constant_tensor": (1, c2, cfg["base_scale_y"], cfg["base_scale_x"]),
        "latent_avg": (512,),  # hard-coded 512 in the reference (networks_stylegan.py:97)
        "truncation_psi": (L,),
    }
    for i in range(8):
        shapes["mp_dense_%d_weight" % i] = (ls, ls)
        shapes["mp_dense_%d_bias" % i] = (ls,)
        if cfg["use_wscale"]:
            shapes["mp_dense_%d_std" % i] = (1,)
    for r in range(2, cfg["max_res_log2"] + 1):
        R = 2 ** r
        C = num_features(cfg, r)
        Cin = num_features(cfg, r - 1) if r > 2 else C
        if r > 2:
            if r >= 7:
                shapes["%d_deconv_1_weight" % R] = (Cin, C, 4, 4)
                if cfg["use_wscale"]:
                    shapes["%d_deconv_1_std" % R] = (1,)
            else:
                shapes["%d_conv_1_weight" % R] = (C, Cin, 3, 3)
                if cfg["use_wscale"]:
                    shapes["%d_conv_1_std" % R] = (1,)
            shapes["%d_blur_1_w_kernel" % R] = (C, 1, 3, 3)
        for k in (1, 2):
            shapes["%d_noise_%d_scale_factors" % (R, k)] = (1, C, 1, 1)
            shapes["%d_bias_%d_bias" % (R, k)] = (1, C, 1, 1)
            if k == 2:
                shapes["%d_conv_2_weight" % R] = (C, C, 3, 3)
                if cfg["use_wscale"]:
                    shapes["%d_conv_2_std" % R] = (1,)
            shapes["%d_adain_%d_dense_affine_weight" % (R, k)] = (2 * C, ls)
            shapes["%d_adain_%d_dense_affine_bias" % (R, k)] = (2 * C,)
            if cfg["use_wscale"]:
                shapes["%d_adain_%d_dense_affine_std" % (R, k)] = (1,)
            shapes["%d_adain_%d_norm_gamma" % (R, k)] = (C,)
            shapes["%d_adain_%d_norm_beta" % (R, k)] = (C,)
    Rm = 2 ** cfg["max_res_log2"]
    Cm = num_features(cfg, cfg["max_res_log2"])
    shapes["%d_conv_to_rgb_weight" % Rm] = (cfg["channels"], Cm, 1, 1)
    shapes["%d_conv_to_rgb_bias" % Rm] = (cfg["channels"],)
    if cfg["use_wscale"]:
        shapes["%d_conv_to_rgb_std" % Rm] = (1,)
    return shapes


def blur_kernel(channels, filter_kernel=(1, 2, 1)):
    # reference networks_stylegan.py:211-228: outer product, normalised, repeated per channel
    f = np.asarray(filter_kernel, dtype=np.float32)
    k = (f[None, :] * f[:, None]).astype(np.float32)
    k = (k / np.float32(k.sum())).astype(np.float32)
    return np.repeat(k[None, None], channels, axis=0).astype(np.float32)


def formula_constants(cfg):
    """The ``std`` / blur constants the constructors compute (used when a file omits them)."""
    out = {}
    ls = cfg["latent_size"]
    g2 = np.sqrt(2)
    for i in range(8):
        out["mp_dense_%d_std" % i] = _std(g2, ls)
    for r in range(2, cfg["max_res_log2"] + 1):
        R = 2 ** r
        C = num_features(cfg, r)
        Cin = num_features(cfg, r - 1) if r > 2 else C
        if r > 2:
            if r >= 7:
                out["%d_deconv_1_std" % R] = _std(g2, 16 * Cin)
            else:
                out["%d_conv_1_std" % R] = _std(g2, 9 * Cin)
            out["%d_blur_1_w_kernel" % R] = blur_kernel(C)
        out["%d_conv_2_std" % R] = _std(g2, 9 * C)
        for k in (1, 2):
            out["%d_adain_%d_dense_affine_std" % (R, k)] = _std(1.0, ls)
    Rm = 2 ** cfg["max_res_log2"]
    Cm = num_features(cfg, cfg["max_res_log2"])
    out["%d_conv_to_rgb_std" % Rm] = _std(1.0, Cm)
    return out


def synthetic_generator_params(cfg, seed=2, trivial_norm=True):
    """Random generator weights.  ``trivial_norm=False`` also
    randomises the InstanceNorm gamma/beta and perturbs the blur taps so that tests
    exercise those loaded constants."""
    rng = np.random.Generator(np.random.PCG64(seed))
    consts = formula_constants(cfg)
    L = num_style_layers(cfg)
    out = {}
    for name, shape in generator_param_shapes(cfg).items():
        if name in consts:
            v = consts[name].copy()
            if name.endswith("w_kernel") and not trivial_norm:
                v = (v * rng.uniform(0.8, 1.2, size=v.shape)).astype(np.float32)
        elif name == "truncation_psi":
            v = np.array([0.7] * min(8, L) + [1.0] * max(0, L - 8), dtype=np.float32)
        elif name == "latent_avg":
            v = rng.normal(0, 0.1, size=shape)
        elif name.endswith("norm_gamma"):
            v = np.ones(shape) if trivial_norm else rng.uniform(0.5, 1.5, size=shape)
        elif name.endswith("norm_beta"):
            v = np.zeros(shape) if trivial_norm else rng.normal(0, 0.1, size=shape)
        elif name.endswith("_weight") or name == "constant_tensor":
            v = rng.normal(0, 1.0, size=shape)
        else:  # biases, noise scale factors
            v = rng.normal(0, 0.1, size=shape)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def decoder_param_shapes(cfg):
    """Ordered ``{scheme-S name: shape}`` of the decoder (reference networks_seg.py:49-94)."""
    F, I = cfg["features"], cfg["in_channels"]
    n = len(I)
    s0 = cfg["start_res"]
    shapes = {}

    def bn(prefix, c):
        for p in ("gamma", "beta", "running_mean", "running_var"):
            shapes["%s.%s" % (prefix, p)] = (c,)

    for i in range(s0, n):
        shapes["cvt_block_%d.0.weight" % i] = (F[i], I[i], 3, 3)
        shapes["cvt_block_%d.0.bias" % i] = (F[i],)
        if cfg["use_bn"]:
            bn("cvt_block_%d.1" % i, F[i])
    for i in range(s0, n):
        cs = F[i + 1]
        in_c = F[i] * (2 if i > s0 else 1)
        if i < n - 1:
            b = "main_block_%d.1.base_layers" % i
            second = 3 if cfg["use_bn"] else 2
            shapes[b + ".0.weight"] = (cs, in_c, 3, 3)
            shapes[b + ".0.bias"] = (cs,)
            if cfg["use_bn"]:
                bn(b + ".1", cs)
            shapes[b + ".%d.weight" % second] = (cs, cs, 3, 3)
            shapes[b + ".%d.bias" % second] = (cs,)
            if cfg["use_bn"]:
                bn(b + ".%d" % (second + 1), cs)
            if cs != in_c:
                shapes["main_block_%d.1.shortcut.0.weight" % i] = (cs, in_c, 1, 1)
                shapes["main_block_%d.1.shortcut.0.bias" % i] = (cs,)
        else:
            shapes["main_block_%d.0.weight" % i] = (cs, in_c, 3, 3)
            shapes["main_block_%d.0.bias" % i] = (cs,)
    return shapes


def initial_decoder_params(cfg, seed=1):
    """Fresh decoder for training, as reference ``init_net`` (seg_solver.py:36-46): conv weights
    ``mx.init.Xavier(factor_type='in', magnitude=2.34)`` = U(-s, s) with s = sqrt(2.34/fan_in), biases 0,
    BatchNorm gamma 1 / beta 0 / running_mean 0 / running_var 1 (gluon defaults)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in decoder_param_shapes(cfg).items():
        if name.endswith(".weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            s = math.sqrt(2.34 / fan_in)
            v = rng.uniform(-s, s, size=shape)
        elif name.endswith((".gamma", ".running_var")):
            v = np.ones(shape)
        else:
            v = np.zeros(shape)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def synthetic_decoder_params(cfg, seed=3):
    """Random decoder weights: Xavier(in, 2.34) convs (reference seg_solver.py:38),
    non-trivial BatchNorm statistics."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in decoder_param_shapes(cfg).items():
        if name.endswith(".weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            s = math.sqrt(2.34 / fan_in)
            v = rng.uniform(-s, s, size=shape)
        elif name.endswith(".gamma") or name.endswith(".running_var"):
            v = rng.uniform(0.5, 1.5, size=shape)
        else:  # bias, beta, running_mean
            v = rng.normal(0, 0.1, size=shape)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


# ---------------------------------------------------------------------------
# scheme S (structural) <-> scheme P (prefix) names of the generator

_S_TO_P = [
    (r"^mapping\.(\d+)\.(weight|bias|std)$",
     lambda m: "mp_dense_%d_%s" % ((int(m.group(1)) - 1) // 2, m.group(2))),
    (r"^net(\d+)\.blur\.w_kernel$", lambda m: "%d_blur_1_w_kernel" % 2 ** int(m.group(1))),
    (r"^net(\d+)\.block1\.0\.scale_factors$", lambda m: "%d_noise_1_scale_factors" % 2 ** int(m.group(1))),
    (r"^net(\d+)\.block1\.1\.bias$", lambda m: "%d_bias_1_bias" % 2 ** int(m.group(1))),
    (r"^net(\d+)\.block2\.0\.(weight|std)$", lambda m: "%d_conv_2_%s" % (2 ** int(m.group(1)), m.group(2))),
    (r"^net(\d+)\.block2\.1\.scale_factors$", lambda m: "%d_noise_2_scale_factors" % 2 ** int(m.group(1))),
    (r"^net(\d+)\.block2\.2\.bias$", lambda m: "%d_bias_2_bias" % 2 ** int(m.group(1))),
    (r"^net(\d+)\.adain(\d)\.affine\.(weight|bias|std)$",
     lambda m: "%d_adain_%s_dense_affine_%s" % (2 ** int(m.group(1)), m.group(2), m.group(3))),
    (r"^net(\d+)\.adain(\d)\.instance\.(gamma|beta)$",
     lambda m: "%d_adain_%s_norm_%s" % (2 ** int(m.group(1)), m.group(2), m.group(3))),
    (r"^to_rgb(\d+)\.0\.(weight|bias|std)$",
     lambda m: "%d_conv_to_rgb_%s" % (2 ** int(m.group(1)), m.group(2))),
]


def generator_names_to_scheme_p(tensors):
    """Accept either naming scheme (a '.' in any key means scheme S) and return scheme P."""
    if not any("." in k for k in tensors):
        return dict(tensors)
    out = {}
    for name, v in tensors.items():
        m = re.match(r"^net(\d+)\.block0\.(weight|std)$", name)
        if m:
            r = int(m.group(1))
            kind = "deconv_1" if r >= 7 else "conv_1"  # reference networks_stylegan.py:154
            out["%d_%s_%s" % (2 ** r, kind, m.group(2))] = v
            continue
        for pat, fn in _S_TO_P:
            m = re.match(pat, name)
            if m:
                out[fn(m)] = v
                break
        else:
            out[name] = v  # constant_tensor, latent_avg, truncation_psi, unknown extras
    return out


def generator_names_to_scheme_s(tensors):
    """Scheme P -> scheme S (what ``save_parameters`` of the reference would write)."""
    out = {}
    for name, v in tensors.items():
        m = re.match(r"^mp_dense_(\d+)_(weight|bias|std)$", name)
        if m:
            out["mapping.%d.%s" % (1 + 2 * int(m.group(1)), m.group(2))] = v
            continue
        m = re.match(r"^(\d+)_(.*)$", name)
        if not m:
            out[name] = v
            continue
        r = int(math.log2(int(m.group(1))))
        rest = m.group(2)
        table = [
            (r"^(?:de)?conv_1_(weight|std)$", "net%d.block0.\\1"),
            (r"^blur_1_w_kernel$", "net%d.blur.w_kernel"),
            (r"^noise_1_scale_factors$", "net%d.block1.0.scale_factors"),
            (r"^bias_1_bias$", "net%d.block1.1.bias"),
            (r"^conv_2_(weight|std)$", "net%d.block2.0.\\1"),
            (r"^noise_2_scale_factors$", "net%d.block2.1.scale_factors"),
            (r"^bias_2_bias$", "net%d.block2.2.bias"),
            (r"^adain_(\d)_dense_affine_(weight|bias|std)$", "net%d.adain\\1.affine.\\2"),
            (r"^adain_(\d)_norm_(gamma|beta)$", "net%d.adain\\1.instance.\\2"),
            (r"^conv_to_rgb_(weight|bias|std)$", "to_rgb%d.0.\\1"),
        ]
        for pat, rep in table:
            if re.match(pat, rest):
                out[re.sub(pat, rep % r, rest)] = v
                break
        else:
            out[name] = v
    return out


def complete_generator_params(cfg, tensors, ignore_extra=True):
    """Validate a loaded generator dict against the declared parameters.

    Mirrors ``load_parameters(..., ignore_extra=True)`` of reference
    image_generator.py:22: every declared parameter must be present (``allow_missing``
    is not set there); extra keys are ignored; file values override the constructor
    constants (``std``, blur ``w_kernel``)."""
    tensors = generator_names_to_scheme_p(tensors)
    shapes = generator_param_shapes(cfg)
    out = {}
    missing = [n for n in shapes if n not in tensors]
    if missing:
        raise KeyError("generator parameters missing from file: %s" % ", ".join(missing[:8]))
    for name, shape in shapes.items():
        v = np.asarray(tensors[name])
        if tuple(v.shape) != tuple(shape):
            raise ValueError("parameter %s has shape %s, expected %s" % (name, v.shape, shape))
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    if not ignore_extra:
        extra = [n for n in tensors if n not in shapes]
        if extra:
            raise KeyError("unexpected parameters in file: %s" % ", ".join(extra[:8]))
    return out


def complete_decoder_params(cfg, tensors):
    shapes = decoder_param_shapes(cfg)
    missing = [n for n in shapes if n not in tensors]
    if missing:
        raise KeyError("decoder parameters missing from file: %s" % ", ".join(missing[:8]))
    extra = [n for n in tensors if n not in shapes]
    if extra:  # reference seg_solver.py:347 loads without ignore_extra
        raise KeyError("unexpected decoder parameters: %s" % ", ".join(extra[:8]))
    out = {}
    for name, shape in shapes.items():
        v = np.asarray(tensors[name])
        if tuple(v.shape) != tuple(shape):
            raise ValueError("parameter %s has shape %s, expected %s" % (name, v.shape, shape))
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def synthetic_inputs(cfg, batch, seed_z=0, seed_noise=1):
    """Latents ``z (B,latent)`` and the 2*(max_res_log2-1) noise planes ``(B,1,R,R)``.

    The reference draws both from MXNet's global RNG (image_generator.py:94,
    networks_stylegan.py:297-300), which is not reproducible without MXNet, so they
    are explicit inputs of this implementation."""
    z = np.random.Generator(np.random.PCG64(seed_z)).normal(
        size=(batch, cfg["latent_size"])).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(seed_noise))
    noise = []
    for r in range(2, cfg["max_res_log2"] + 1):
        for _ in range(2):
            noise.append(rng.normal(size=(batch, 1, 2 ** r, 2 ** r)).astype(np.float32))
    return z, noise
