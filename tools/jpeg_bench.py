"""Time the on-device JPEG encoder (csrc/gsa_jpeg.hip) on a batch of GAN-like images: per-kernel HIP-event times
are not split here -- run under `rocprofv3 --kernel-trace --stats` for that."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from gan_segmentation_amd.jpeg import DEFAULT_RESTART, JpegEncoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--restart", type=int, default=DEFAULT_RESTART)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((a.batch, 3, a.res // 16, a.res // 16), device="cuda", generator=g)
    img = torch.nn.functional.interpolate(x, size=(a.res, a.res), mode="bicubic")
    img = (img * 60 + 128 + torch.randn(img.shape, device="cuda", generator=g) * 5).clamp(0, 255).to(torch.uint8)
    img = img.permute(0, 2, 3, 1).contiguous()
    enc = JpegEncoder(a.batch, a.res, a.res, "cuda:0", restart=a.restart)
    for _ in range(3):
        _scan, lengths = enc.encode(img)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.iters):
        enc.encode(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.iters
    ln = lengths.cpu().numpy()
    print("jpeg: res %d batch %d restart %d: %.3f ms per batch, %.0f images/s, %.0f bytes/image (%.1f %% of the pixels)"
          % (a.res, a.batch, a.restart, dt * 1e3, a.batch / dt, ln.mean() + len(enc.header),
             100.0 * ln.mean() / (a.res * a.res * 3)))
    # the mask compressor on blob masks of the same size
    from gan_segmentation_amd.png import PngEncoder
    m = torch.nn.functional.interpolate(torch.randn((a.batch, 1, 8, 8), device="cuda", generator=g), size=(a.res, a.res), mode="bicubic")
    mask = (m[:, 0] > 0).to(torch.uint8).contiguous()
    penc = PngEncoder(a.batch, a.res, a.res, "cuda:0")
    for _ in range(3):
        _s, plen = penc.encode(mask)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.iters):
        penc.encode(mask)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.iters
    print("png:  res %d batch %d: %.3f ms per batch, %.0f masks/s, %.0f bytes/mask (%.2f %% of the pixels)"
          % (a.res, a.batch, dt * 1e3, a.batch / dt, plen.cpu().numpy().mean() + 57, 100.0 * plen.cpu().numpy().mean() / (a.res * a.res)))


if __name__ == "__main__":
    main()
