"""GPU parity tests: the HIP path (through the C ABI) against the canonical C oracle.

The bar is BIT-EXACT equality for every output (fp32 rgb, features, logits, u8 image, u8 mask):
the kernels implement the oracle's canonical fp32 evaluation order."""
import numpy as np
import pytest

from tests.common import reduced_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _first_diff(a, b):
    idx = np.argwhere(a != b)
    return "%d of %d differ, first at %s: %r vs %r, max abs %g" % (
        len(idx), a.size, tuple(idx[0]), a[tuple(idx[0])], b[tuple(idx[0])],
        np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


def assert_same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    assert np.array_equal(a, b), "%s: %s" % (what, _first_diff(a, b))


def test_reduced_generator_decoder_bit_exact(torch_cuda, oracle_lib):
    from gan_segmentation_amd.networks_seg import Decoder
    from gan_segmentation_amd.networks_stylegan import Generator
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=3)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    rgb_o, img_o, feats_o = o.generator(z, noise)
    logits_o, mask_o = o.decoder(feats_o)

    g = Generator(gcfg)
    g.load_parameters(gp)
    d = Decoder(dcfg, 1)
    d.load_parameters(dp)
    rgb, feats, img = g(z, noise=noise, want_image=True)
    for i, (a, b) in enumerate(zip(feats, feats_o)):
        assert_same(a.cpu().numpy(), b, "feature %d" % i)
    assert_same(rgb.cpu().numpy(), rgb_o, "rgb")
    assert_same(img.cpu().numpy(), img_o, "image")
    logits, mask = d(*feats, want_mask=True)
    assert_same(logits.cpu().numpy(), logits_o, "logits")
    assert_same(mask.cpu().numpy(), mask_o, "mask")
