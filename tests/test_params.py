"""`.params` container (MXNet nd.save list format) and name schemes."""
import struct

import numpy as np
import pytest

from gan_segmentation_amd import params as P
from gan_segmentation_amd import weights as W


def handmade_bytes():
    """Hand-assembled file: two arrays, V2 and V3 magics, with an 'arg:' prefixed key."""
    out = struct.pack("<QQQ", 0x112, 0, 2)
    a = np.arange(6, dtype="<f4").reshape(2, 3)
    out += struct.pack("<IiI", 0xF993FAC9, 0, 2) + struct.pack("<2q", 2, 3) + struct.pack("<iii", 1, 0, 0) + a.tobytes()
    b = np.array([7, 8, 9], dtype="<i4")
    out += struct.pack("<IiI", 0xF993FACA, 0, 1) + struct.pack("<1q", 3) + struct.pack("<iii", 1, 0, 4) + b.tobytes()
    out += struct.pack("<Q", 2)
    for name in (b"arg:first_weight", b"second"):
        out += struct.pack("<Q", len(name)) + name
    return out, a, b


def test_handmade_fixture_parses(tmp_path):
    buf, a, b = handmade_bytes()
    got = P.loads_params(buf)
    assert list(got) == ["first_weight", "second"]
    assert np.array_equal(got["first_weight"], a) and got["first_weight"].dtype == np.float32
    assert np.array_equal(got["second"], b) and got["second"].dtype == np.int32
    # the committed copy of the same bytes
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "handmade.params")
    with open(path, "rb") as f:
        assert f.read() == buf


def test_writer_matches_handmade_layout():
    buf, a, _b = handmade_bytes()
    one = P.dumps_params({"k": a})
    assert one[:24] == struct.pack("<QQQ", 0x112, 0, 1)
    assert one[24:36] == struct.pack("<IiI", 0xF993FAC9, 0, 2)
    assert P.loads_params(one)["k"].tolist() == a.tolist()


def test_round_trip_generator_and_decoder(tmp_path):
    gcfg = W.reduced_generator_config(7)
    gp = W.synthetic_generator_params(gcfg)
    path = str(tmp_path / "g.params")
    P.save_params(path, gp)
    back = P.load_params(path)
    assert list(back) == list(gp)
    assert all(np.array_equal(back[k], gp[k]) for k in gp)
    dcfg = W.decoder_config(7, in_channels=W.generator_channels(gcfg))
    dp = W.synthetic_decoder_params(dcfg)
    P.save_params(path, dp, magic=P.NDARRAY_V3_MAGIC)
    back = P.load_params(path)
    assert all(np.array_equal(back[k], dp[k]) for k in dp)


@pytest.mark.parametrize("cut", [3, 20, 40, 70])
def test_truncated_and_bad_magic(cut):
    buf, _a, _b = handmade_bytes()
    with pytest.raises(P.ParamsFormatError):
        P.loads_params(buf[:cut])
    with pytest.raises(P.ParamsFormatError):
        P.loads_params(b"\x13" + buf[1:])


def test_name_schemes_round_trip():
    gcfg = W.generator_config(10)
    shapes = W.generator_param_shapes(gcfg)
    dummy = {k: np.zeros(1, np.float32) for k in shapes}
    s = W.generator_names_to_scheme_s(dummy)
    assert "net10.block0.weight" in s and "mapping.15.weight" in s and "to_rgb10.0.bias" in s
    assert "net2.block2.0.std" in s and "net7.adain2.instance.gamma" in s
    assert set(W.generator_names_to_scheme_p(s)) == set(shapes)


def test_ffhq_shapes_match_survey_appendix_b():
    gcfg = W.generator_config(10)
    sh = W.generator_param_shapes(gcfg)
    assert W.generator_channels(gcfg) == [512, 512, 512, 512, 256, 128, 64, 32, 16]
    assert sh["constant_tensor"] == (1, 512, 4, 4) and sh["truncation_psi"] == (18,)
    assert sh["64_conv_1_weight"] == (256, 512, 3, 3) and sh["128_deconv_1_weight"] == (256, 128, 4, 4)
    assert sh["1024_conv_to_rgb_weight"] == (3, 16, 1, 1) and sh["8_blur_1_w_kernel"] == (512, 1, 3, 3)
    assert sh["1024_adain_2_dense_affine_weight"] == (32, 512)
    assert "4_conv_1_weight" not in sh and "4_blur_1_w_kernel" not in sh
    n_params = sum(int(np.prod(s)) for s in sh.values())
    assert 26.0e6 < n_params < 27.0e6          # ~26.5 M
    dsh = W.decoder_param_shapes(W.decoder_config(10))
    assert dsh["cvt_block_8.0.weight"] == (16, 16, 3, 3)
    assert dsh["main_block_7.1.shortcut.0.weight"] == (16, 64, 1, 1)
    assert dsh["main_block_8.0.weight"] == (2, 32, 3, 3)
    assert "main_block_0.1.shortcut.0.weight" not in dsh
    assert 0.9e6 < sum(int(np.prod(s)) for s in dsh.values()) < 1.0e6


def test_missing_and_extra_params():
    gcfg = W.reduced_generator_config(7)
    gp = W.synthetic_generator_params(gcfg)
    W.complete_generator_params(gcfg, dict(gp, lod=np.zeros(1)))          # extras ignored
    with pytest.raises(KeyError):
        W.complete_generator_params(gcfg, {k: v for k, v in gp.items() if k != "latent_avg"})
    with pytest.raises(ValueError):
        W.complete_generator_params(gcfg, dict(gp, latent_avg=np.zeros(7, np.float32)))
    dcfg = W.decoder_config(7, in_channels=W.generator_channels(gcfg))
    with pytest.raises(KeyError):
        W.complete_decoder_params(dcfg, dict(W.synthetic_decoder_params(dcfg), bogus=np.zeros(1)))
