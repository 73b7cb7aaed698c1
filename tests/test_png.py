"""The dataset writer's mask compressor (reference main.py:102-103 cv2.imwrite(.png)):
* CPU: the Python restatement (oracle/ref_png.py) is pinned by zlib itself -- inflating its stream gives the filtered
  scanlines, a PNG reader gives the mask back; the C ABI of include/gsa_png.h against the library exports;
* GPU (-m gpu): csrc/gsa_png.hip byte for byte against the restatement on small masks, decoded back on full sizes."""
import ctypes
import io
import os
import re
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def masks(rng, H, W):
    blob = np.zeros((H, W), np.uint8)
    blob[H // 5:H // 2 + 3, W // 4:W - W // 3] = 1
    blob[H // 2:, : W // 8] = 2
    return {"binary noise": rng.integers(0, 2, (H, W), dtype=np.uint8), "8 classes": rng.integers(0, 8, (H, W), dtype=np.uint8),
            "all bytes": rng.integers(0, 256, (H, W), dtype=np.uint8), "zeros": np.zeros((H, W), np.uint8),
            "constant 7": np.full((H, W), 7, np.uint8), "blobs": blob}


def decode(png_bytes):
    from PIL import Image
    im = Image.open(io.BytesIO(png_bytes))
    im.load()                                   # raises on a bad CRC / Adler-32 / deflate stream
    assert im.mode == "L"
    return np.asarray(im)


@pytest.mark.parametrize("H,W", [(1, 16), (4, 16), (5, 32), (16, 16), (33, 48), (64, 64), (9, 1040)])
def test_restatement_is_a_valid_zlib_stream(H, W):
    from gan_segmentation_amd.png import png_file
    from oracle import ref_png as R
    rng = np.random.default_rng(H * 7 + W)
    for kind, m in masks(rng, H, W).items():
        s = R.zlib_stream(m)
        assert zlib.decompress(s) == R.filtered_scanlines(m).tobytes(), kind
        assert np.array_equal(decode(png_file(H, W, s)), m), kind


def test_png_header_symbols_are_exported(hip_library):
    with open(os.path.join(ROOT, "include", "gsa_png.h")) as f:
        text = f.read()
    declared = set(re.findall(r"\b(gsa_png_[a-z0-9_]+)\s*\(", text))
    assert declared == {"gsa_png_workspace_bytes", "gsa_png_max_stream_bytes", "gsa_png_encode"}
    lib = ctypes.CDLL(hip_library)
    for name in declared:
        assert hasattr(lib, name), "%s declared in gsa_png.h but not exported" % name
    from gan_segmentation_amd import png
    api = png._api()
    assert set(api.keys()) == declared
    assert api["gsa_png_workspace_bytes"](1, 64, 60) < 0 and api["gsa_png_max_stream_bytes"](64, 8) < 0     # W % 16
    ws = api["gsa_png_workspace_bytes"](2, 64, 64)
    assert ws > 0
    good = dict(n=2, H=64, W=64, mask=4096, ws=8192, wsb=ws, out=1 << 20, stride=1 << 16, ln=1 << 22)

    def call(**kw):          # argument validation happens on the host, before any HIP call
        a = dict(good, **kw)
        return api["gsa_png_encode"](None, a["n"], a["H"], a["W"], a["mask"], a["ws"], a["wsb"], a["out"], a["stride"], a["ln"])

    for bad in (dict(n=0), dict(W=40), dict(H=0), dict(mask=None), dict(mask=4100), dict(ws=None), dict(wsb=ws - 1), dict(out=None),
                dict(ln=None), dict(stride=4)):
        assert call(**bad) == -1, bad


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,n", [(1, 16, 1), (5, 32, 3), (16, 16, 2), (33, 48, 4), (64, 64, 6)])
def test_hip_stream_is_byte_identical_to_the_restatement(torch_cuda, H, W, n):
    import torch
    from gan_segmentation_amd.png import PngEncoder, png_file
    from oracle import ref_png as R
    rng = np.random.default_rng(H + W)
    ms = masks(rng, H, W)
    kinds = list(ms)
    enc = PngEncoder(n, H, W, "cuda:0")
    for first in range(0, len(kinds), n):
        batch = [ms[k] for k in kinds[first:first + n]]
        files = enc.files(torch.from_numpy(np.stack(batch)).cuda())
        for kind, m, f in zip(kinds[first:], batch, files):
            assert f == png_file(H, W, R.zlib_stream(m)), kind


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,n", [(1024, 1024, 3), (256, 512, 6), (7, 2048, 2)])
def test_hip_png_decodes_back(torch_cuda, H, W, n):
    import torch
    from gan_segmentation_amd.png import PngEncoder
    rng = np.random.default_rng(H * 3 + W)
    ms = masks(rng, H, W)
    kinds = list(ms)
    enc = PngEncoder(n, H, W, "cuda:0")
    sizes = {}
    for first in range(0, len(kinds), n):
        batch = [ms[k] for k in kinds[first:first + n]]
        for kind, m, f in zip(kinds[first:], batch, enc.files(torch.from_numpy(np.stack(batch)).cuda())):
            assert np.array_equal(decode(f), m), kind
            sizes[kind] = len(f)
    assert sizes["blobs"] < H * W // 50 + 256 and sizes["zeros"] < H * W // 64 + 256        # ~1 % of the pixels


@pytest.mark.gpu
def test_dataset_writer_with_gpu_jpeg_and_png(torch_cuda, tmp_path):
    """What `main.py generate` does by default: both files of a pair compressed on the GPU."""
    import torch
    from PIL import Image
    from gan_segmentation_amd.dataset_writer import DatasetWriter
    rng = np.random.default_rng(9)
    n, R = 10, 128
    img = rng.integers(0, 256, (n, R, R, 3), dtype=np.uint8)
    img[:5] = (np.linspace(0, 255, R)[None, None, :, None] * np.ones((5, R, 1, 3))).astype(np.uint8)
    mk = np.stack([m for m in list(masks(rng, R, R).values())[:5]] * 2)
    dimg, dmask = torch.from_numpy(img).cuda(), torch.from_numpy(mk).cuda()
    for gpu_jpeg in (True, False):
        d = tmp_path / ("jpeg%d" % gpu_jpeg)
        with DatasetWriter(str(d), workers=4, gpu_jpeg=gpu_jpeg, gpu_png=True) as w:
            w.submit(dimg[:4], dmask[:4], 0)
            w.submit(dimg[4:8], dmask[4:8], 4)
            w.submit(dimg[8:], dmask[8:], 8)
        assert w.written == n
        assert sorted(os.listdir(d)) == sorted(["img_%06d.jpg" % i for i in range(n)] + ["mask_%06d.png" % i for i in range(n)])
        for i in range(n):
            assert np.array_equal(np.asarray(Image.open(d / ("mask_%06d.png" % i))), mk[i]), i
            b = io.BytesIO()
            Image.fromarray(img[i], "RGB").save(b, "JPEG", quality=95)
            ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
            assert np.array_equal(np.asarray(Image.open(d / ("img_%06d.jpg" % i)).convert("RGB")), ref), i
