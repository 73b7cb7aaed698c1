"""The dataset writer's JPEG encoder (reference main.py:100-101 cv2.imwrite -> libjpeg-turbo):
* CPU: the scalar oracle is PINNED byte for byte against libjpeg-turbo (through Pillow, the same library cv2 links),
  the product library's header function against both, the C ABI of include/gsa_jpeg.h against the exports;
* GPU (-m gpu): csrc/gsa_jpeg.hip byte for byte against the oracle, and through the dataset writer."""
import ctypes
import io
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def images(rng, H, W):
    """-> {kind: (H,W,3) u8}: a smooth GAN-like picture, white noise (longest codes, many 0xFF bytes), flat black /
    white (empty blocks, DC only), saturated primaries in stripes (largest chroma swings)."""
    import torch
    import torch.nn.functional as F
    x = rng.standard_normal((1, 3, max(H // 16, 2), max(W // 16, 2))).astype(np.float32)
    smooth = F.interpolate(torch.from_numpy(x), size=(H, W), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    smooth = (smooth * 60 + 128 + rng.standard_normal((H, W, 3)) * 5).clip(0, 255).astype(np.uint8)
    stripes = np.zeros((H, W, 3), np.uint8)
    for i in range(W):
        stripes[:, i, (i // 3) % 3] = 255 if (i // 5) % 2 else 0
    stripes[::7] = 255 - stripes[::7]
    return {"smooth": np.ascontiguousarray(smooth), "noise": rng.integers(0, 256, (H, W, 3), dtype=np.uint8),
            "black": np.zeros((H, W, 3), np.uint8), "white": np.full((H, W, 3), 255, np.uint8), "stripes": stripes}


def pillow_bytes(img, quality, restart):
    from PIL import Image
    b = io.BytesIO()
    kw = {"restart_marker_blocks": restart} if restart else {}
    Image.fromarray(img, "RGB").save(b, "JPEG", quality=quality, **kw)
    return b.getvalue()


@pytest.mark.parametrize("H,W,quality,restart", [(16, 16, 95, 0), (32, 48, 95, 1), (64, 64, 95, 4), (128, 256, 95, 3),
                                                 (256, 256, 75, 0), (256, 128, 100, 5), (64, 64, 10, 2), (48, 32, 50, 7),
                                                 (512, 512, 95, 4)])
def test_oracle_is_byte_identical_to_libjpeg_turbo(H, W, quality, restart):
    """Pins oracle/c/jpeg_oracle.c: whole files (headers, tables, scan, restart markers) equal Pillow's libjpeg-turbo
    output -- the library behind the reference's cv2.imwrite -- for the same quality and restart interval."""
    from PIL import features
    if not features.check("jpg"):
        pytest.skip("Pillow without JPEG support")
    from oracle import jpeg_binding as J
    rng = np.random.default_rng(H * 131 + W + quality)
    for kind, img in images(rng, H, W).items():
        ours, ref = J.encode(img, quality, restart), pillow_bytes(img, quality, restart)
        assert ours == ref, "%s %dx%d q%d ri%d: %d vs %d bytes" % (kind, H, W, quality, restart, len(ours), len(ref))


def test_oracle_reproduces_the_committed_libjpeg_turbo_files():
    """The same pin without Pillow: files libjpeg-turbo wrote (tests/golden/make_jpeg_golden.py) for three images at
    three quality / restart settings."""
    from oracle import jpeg_binding as J
    g = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_libjpeg_turbo.npz"))
    keys = [k[:-4] for k in g.files if k.endswith("_rgb")]
    assert len(keys) == 9
    for k in keys:
        quality, restart = int(k.split("_q")[1].split("_")[0]), int(k.split("_ri")[1])
        assert J.encode(g[k + "_rgb"], quality, restart) == g[k + "_jpg"].tobytes(), k


def test_header_of_the_product_library(hip_library):
    """gsa_jpeg_header (host function of the HIP library) == the oracle's == the front of Pillow's file."""
    from oracle import jpeg_binding as J
    lib = ctypes.CDLL(hip_library)
    lib.gsa_jpeg_header.restype = ctypes.c_int64
    lib.gsa_jpeg_header.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_void_p, ctypes.c_int64]
    for H, W, q, ri in [(1024, 1024, 95, 4), (256, 512, 75, 1), (16, 16, 100, 0), (512, 512, 30, 65535)]:
        buf = ctypes.create_string_buffer(1024)
        n = lib.gsa_jpeg_header(H, W, q, ri, ctypes.cast(buf, ctypes.c_void_p), 1024)
        assert n == (629 if ri else 623)
        assert buf.raw[:n] == J.header(H, W, q, ri)
        if H * W <= 256 * 512:
            assert pillow_bytes(np.zeros((H, W, 3), np.uint8), q, ri)[:n] == buf.raw[:n]
    assert lib.gsa_jpeg_header(70000, 16, 95, 4, None, 0) < 0
    assert lib.gsa_jpeg_header(1024, 1024, 95, 4, None, 0) == 629            # sizing call


def test_jpeg_header_symbols_are_exported(hip_library):
    with open(os.path.join(ROOT, "include", "gsa_jpeg.h")) as f:
        text = f.read()
    declared = set(re.findall(r"\b(gsa_jpeg_[a-z0-9_]+)\s*\(", text))
    assert declared == {"gsa_jpeg_header", "gsa_jpeg_workspace_bytes", "gsa_jpeg_max_scan_bytes", "gsa_jpeg_encode"}
    lib = ctypes.CDLL(hip_library)
    for name in declared:
        assert hasattr(lib, name), "%s declared in gsa_jpeg.h but not exported" % name
    from gan_segmentation_amd import jpeg
    assert set(jpeg._api().keys()) == declared
    # argument checks that need no GPU
    assert jpeg._api()["gsa_jpeg_workspace_bytes"](1, 100, 64, 4) < 0          # not a multiple of 16
    assert jpeg._api()["gsa_jpeg_workspace_bytes"](1, 64, 64, 0) < 0           # restart interval required
    assert jpeg._api()["gsa_jpeg_max_scan_bytes"](1024, 1024, 4) == 4096 * 6 * 448 + 1024 * 3 + 2


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,quality,restart,n", [(16, 16, 95, 1, 1), (64, 48, 95, 4, 3), (128, 256, 75, 3, 2),
                                                   (256, 256, 100, 5, 2), (64, 64, 10, 2, 5), (1024, 1024, 95, 4, 2),
                                                   (512, 512, 95, 1024, 1)])
def test_hip_encoder_is_byte_identical_to_the_oracle(torch_cuda, H, W, quality, restart, n):
    import torch
    from gan_segmentation_amd.jpeg import JpegEncoder
    from oracle import jpeg_binding as J
    rng = np.random.default_rng(7 + H + W + quality)
    pics = images(rng, H, W)
    kinds = list(pics)
    enc = JpegEncoder(n, H, W, "cuda:0", quality=quality, restart=restart)
    for first in range(0, len(kinds), n):
        batch = [pics[k] for k in kinds[first:first + n]]
        files = enc.files(torch.from_numpy(np.stack(batch)).cuda())
        for kind, img, f in zip(kinds[first:], batch, files):
            assert f == J.encode(img, quality, restart), "%s %dx%d q%d ri%d" % (kind, H, W, quality, restart)
    # a decoder accepts it, and it is what libjpeg-turbo writes for these pixels
    from PIL import Image
    dec = np.asarray(Image.open(io.BytesIO(files[-1])).convert("RGB"))
    ref = np.asarray(Image.open(io.BytesIO(pillow_bytes(batch[-1], quality, restart))).convert("RGB"))
    assert np.array_equal(dec, ref)


@pytest.mark.gpu
def test_hip_encoder_reports_a_short_buffer(torch_cuda):
    import torch
    from gan_segmentation_amd.jpeg import JpegEncoder
    from oracle import jpeg_binding as J
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (1, 64, 64, 3), dtype=np.uint8)
    enc = JpegEncoder(1, 64, 64, "cuda:0", quality=100, restart=2, out_stride=1000)
    _scan, lengths = enc.encode(torch.from_numpy(img).cuda())
    need = len(J.encode(img[0], 100, 2)) - len(enc.header)
    assert int(lengths.cpu()[0]) == -need
    assert enc.files(torch.from_numpy(img).cuda())[0] == J.encode(img[0], 100, 2)      # retried with the worst-case stride


@pytest.mark.gpu
def test_dataset_writer_with_gpu_jpeg(torch_cuda, tmp_path):
    """DatasetWriter(gpu_jpeg=True): the files `main.py generate` writes -- names as reference main.py:100-103, the
    JPEG decodes to exactly what libjpeg-turbo (cv2) stores for the same pixels, the PNG holds the class indices."""
    import torch
    from PIL import Image
    from gan_segmentation_amd.dataset_writer import DatasetWriter
    rng = np.random.default_rng(5)
    n, R = 11, 128
    pics = images(rng, R, R)
    img = np.stack([pics[k] for k in ("smooth", "noise", "stripes", "black", "white", "smooth", "smooth", "noise", "stripes", "smooth", "noise")])
    mask = (rng.random((n, R, R)) > 0.5).astype(np.uint8)
    dimg, dmask = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
    with DatasetWriter(str(tmp_path), workers=4, gpu_jpeg=True) as w:
        w.submit(dimg[:4], dmask[:4], 0)
        w.submit(dimg[4:8], dmask[4:8], 4)
        w.submit(dimg[8:], dmask[8:], 8)       # a short last batch
    assert w.written == n
    assert sorted(os.listdir(tmp_path)) == sorted(["img_%06d.jpg" % i for i in range(n)] + ["mask_%06d.png" % i for i in range(n)])
    for i in range(n):
        got = np.asarray(Image.open(tmp_path / ("img_%06d.jpg" % i)).convert("RGB"))
        ref = np.asarray(Image.open(io.BytesIO(pillow_bytes(img[i], 95, 0))).convert("RGB"))
        assert np.array_equal(got, ref), i
        assert np.array_equal(np.asarray(Image.open(tmp_path / ("mask_%06d.png" % i))), mask[i])


def test_encode_rejects_bad_arguments_before_touching_the_gpu(hip_library):
    """Argument validation of gsa_jpeg_encode happens on the host (no HIP call precedes it): sizes that are not
    multiples of 16, a missing restart interval, null / misaligned pointers, a workspace that is too small."""
    from gan_segmentation_amd import jpeg
    enc = jpeg._api()["gsa_jpeg_encode"]
    ws = jpeg._api()["gsa_jpeg_workspace_bytes"](1, 64, 64, 2)
    good = dict(n=1, H=64, W=64, rgb=4096, q=95, ri=2, ws=8192, wsb=ws, out=16384, stride=1 << 20, ln=32768)

    def call(**kw):
        a = dict(good, **kw)
        return enc(None, a["n"], a["H"], a["W"], a["rgb"], a["q"], a["ri"], a["ws"], a["wsb"], a["out"], a["stride"], a["ln"])

    for bad in (dict(n=0), dict(H=60), dict(W=8), dict(ri=0), dict(ri=70000), dict(rgb=None), dict(rgb=4097), dict(ws=None),
                dict(wsb=ws - 1), dict(out=None), dict(ln=None), dict(stride=1), dict(H=65536 + 16)):
        assert call(**bad) == -1, bad
