"""GPU tests of the bf16-MFMA mode (BASELINE.json configs[4]: "bf16 MFMA", fp32 accumulate and statistics).

Tolerance contract (stated here, loosened from the fp32 path's bit equality / 1e-3):
 * the arithmetic differs from fp32 in the operands of the MFMA convolutions (inputs after AdaIN and weights rounded to
   bf16, 8 significant bits) and -- round 2 -- in the activation tensors that live in HBM, which are bf16 too (the producer
   rounds what it stores after taking the statistics from the fp32 values; the oracle's bf16 mode rounds at the same
   points); every arithmetic step in between is the canonical fp32 code;
 * against the C oracle run in the same mode (oracle/c/gsa_oracle.c, bf16r()) the first synthesis level must
   agree to fp32 rounding (<= 2e-6 of the tensor's range): same operands, only the matrix core's internal
   summation order differs (tools/probe/).  Deeper levels amplify single flipped bf16 roundings, so the
   end-to-end bars are: max|d rgb| <= 3 % and mean|d rgb| <= 0.3 % of the rgb range, masks agree on >= 99.5 %
   of the pixels;
 * against the fp32 path: max <= 6 %, mean <= 0.6 % of the range, masks agree on >= 99 % of the pixels.
The synthetic weights put rgb in about [-7, 7]; "range" is max|reference|.
"""
import numpy as np
import pytest

from tests.common import gan_setup, reduced_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _build(setup, batch, precision):
    from gan_segmentation_amd.image_generator import ImageGenerator
    gcfg, gp, dcfg, dp, _z, _noise = setup
    return ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=batch, precision=precision)


def _rel(a, b):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    scale = float(np.abs(b).max())
    return d.max() / scale, d.mean() / scale


def _check_against(rgb, mask, rgb_ref, mask_ref, max_rel, mean_rel, min_agree, what):
    mx, mean = _rel(rgb, rgb_ref)
    agree = float(np.mean(np.asarray(mask) == np.asarray(mask_ref)))
    assert mx <= max_rel and mean <= mean_rel, "%s: rgb max %.3e mean %.3e of range" % (what, mx, mean)
    assert agree >= min_agree, "%s: masks agree on %.5f of the pixels" % (what, agree)


def _run(gen, z, noise):
    rgb, feats, img = gen.netG(z, noise=noise, want_image=True)
    logits, mask = gen._decoder(*feats, want_mask=True)
    return rgb.cpu().numpy(), [f.cpu().numpy() for f in feats], img.cpu().numpy(), mask.cpu().numpy()


def test_bf16_reduced_against_bf16_oracle_and_fp32(torch_cuda, oracle_lib):
    setup = reduced_setup(7, batch=3)
    gcfg, gp, dcfg, dp, z, noise = setup
    rgb, feats, img, mask = _run(_build(setup, 3, "bf16"), z, noise)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp, precision="bf16")
    rgb_o, _img_o, feats_o = o.generator(z, noise)
    _logits_o, mask_o = o.decoder(feats_o)
    mx, _ = _rel(feats[0], feats_o[0])
    assert mx <= 2e-6, "4x4 level must agree with the bf16 oracle to fp32 rounding, got %.3e" % mx
    _check_against(rgb, mask, rgb_o, mask_o, 2e-2, 2e-3, 0.997, "bf16 HIP vs bf16 oracle")       # measured 0.4 % / 0.011 % / 99.97 %
    # the mode really is a different arithmetic, and stays close to the canonical fp32 path
    rgb32, _f32, img32, mask32 = _run(_build(setup, 3, "fp32"), z, noise)
    assert not np.array_equal(rgb, rgb32)
    _check_against(rgb, mask, rgb32, mask32, 6e-2, 6e-3, 0.99, "bf16 HIP vs fp32 HIP")                # measured 4.8 % / 0.29 % / 99.72 % (128 px, 3 samples)
    o32 = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    assert np.array_equal(o32.generator(z, noise)[0], rgb32), "the fp32 context next to a bf16 one stays bit-exact"


def test_bf16_fused_generate_equals_two_calls(torch_cuda):
    setup = reduced_setup(7, batch=5, trivial_norm=False)
    _gcfg, _gp, _dcfg, _dp, z, noise = setup
    gen = _build(setup, 5, "bf16")
    img, mask = gen.generate_batch(z, noise)
    _rgb, _feats, img2, mask2 = _run(gen, z, noise)
    assert np.array_equal(img.cpu().numpy(), img2) and np.array_equal(mask.cpu().numpy(), mask2)
    # batch composition still does not change a sample (sharding property)
    img_a, mask_a = gen.generate_batch(z[3:], [a[3:] for a in noise])
    assert np.array_equal(img_a.cpu().numpy(), img2[3:]) and np.array_equal(mask_a.cpu().numpy(), mask2[3:])


def test_precision_is_fixed_once_weights_are_committed(torch_cuda):
    from gan_segmentation_amd import _lib
    gen = _build(reduced_setup(7, batch=1), 1, "fp32")
    ctx = gen.netG._model.ctx
    with pytest.raises(_lib.GsaError, match="gsa_set_precision must precede"):
        ctx.set_precision("bf16")
    ctx.set_precision("fp32")          # re-stating the current mode is fine
    with pytest.raises(KeyError):
        ctx.set_precision("fp8")
    with pytest.raises(_lib.GsaError, match="precision must be one of"):
        _build(reduced_setup(7, batch=1), 1, "fp16")


def test_config5_cars_512_bf16(torch_cuda, oracle_lib):
    """BASELINE.json configs[4]: stylegan-cars 512^2 synthesis + decoder, bf16 MFMA, 4 samples per GPU."""
    setup = gan_setup("cars", batch=4)
    gcfg, gp, dcfg, dp, z, noise = setup
    gen = _build(setup, 4, "bf16")
    img, mask = gen.generate_batch(z, noise)
    rgb, _feats, img2, mask2 = _run(gen, z, noise)
    assert np.array_equal(img.cpu().numpy(), img2) and np.array_equal(mask.cpu().numpy(), mask2)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp, precision="bf16")
    rgb_o, img_o, feats_o = o.generator(z[:1], [a[:1] for a in noise])
    _logits_o, mask_o = o.decoder(feats_o)
    _check_against(rgb[:1], mask2[:1], rgb_o, mask_o, 2e-2, 2e-3, 0.997, "cars bf16 HIP vs bf16 oracle")   # measured 1.4 % / 0.13 % / 99.80 %
    assert np.abs(img2[:1].astype(np.int32) - img_o.astype(np.int32)).mean() <= 2.0   # u8 image: mean error <= 2 levels
    rgb32, _f, img32, mask32 = _run(_build(setup, 4, "fp32"), z, noise)
    _check_against(rgb, mask2, rgb32, mask32, 4.5e-2, 4e-3, 0.993, "cars bf16 HIP vs fp32 HIP")          # measured 3.3 % / 0.25 % / 99.57 %
    # the fp32 path at this batch (4096 tiles at 512^2: the persistent resident-weight kernels are in use) stays bit-exact
    o32 = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    img_o32, mask_o32 = o32.generate(z[3:], [a[3:] for a in noise])
    assert np.array_equal(img32[3:], img_o32) and np.array_equal(mask32[3:], mask_o32)


_BF16_LEAN_WORKER = r'''
import hashlib, sys
sys.path.insert(0, ROOT_DIR)
import numpy as np
from tests.common import gan_setup, reduced_setup
from gan_segmentation_amd.image_generator import ImageGenerator
for name, setup, batch in (("reduced", reduced_setup(7, batch=3, trivial_norm=False), 3), ("cars", gan_setup("cars", 2), 2)):
    gcfg, gp, dcfg, dp, z, noise = setup
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=batch, precision="bf16")
    rgb, feats, img = gen.netG(z, noise=noise, want_image=True)
    logits, mask = gen._decoder(*feats, want_mask=True)
    h = hashlib.sha256()
    for t in [rgb, img, logits, mask] + list(feats):
        h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
    print("DIGEST", name, h.hexdigest())
'''


def test_bf16_lean_kernels_give_the_general_kernels_values(torch_cuda, tmp_path):
    """conv3x3_bf16_lean (round 5) replaces conv3x3_mfma<..., BF = true> from 32 px on: the same products in the same chain order with the same
    roundings -- not a tolerance: every tensor of the bf16 mode (rgb, image, logits, mask, all features) has the same bytes with GSA_BF16_LEAN=0
    and =1, on the reduced 128-px model (odd batch, loaded norm parameters) and on cars 512^2.  (Child processes: the switch is read once.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "bf16_lean_worker.py"
    script.write_text(_BF16_LEAN_WORKER.replace("ROOT_DIR", repr(root)))
    digests = {}
    for v in ("0", "1"):
        out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, GSA_BF16_LEAN=v, GSA_GRAPH="0"), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-800:] + out.stderr[-2500:]
        digests[v] = sorted(l for l in out.stdout.splitlines() if l.startswith("DIGEST"))
        assert len(digests[v]) == 2, out.stdout[-800:]
    assert digests["0"] == digests["1"], digests
