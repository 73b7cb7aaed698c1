#!/usr/bin/env python3
"""Throughput of the dataset writer alone on GAN-like content resident in HBM -- smooth pictures with
fine noise (a q95 JPEG of ~15 % of the pixels) and blob masks -- with the JPEG encoded by the host pool (PIL /
libjpeg-turbo) or by the HIP kernels:   python tools/writer_bench.py [res=1024] [pairs=1024] [batch=8] [workers]"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_segmentation_amd.dataset_writer import DatasetWriter, default_workers  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
workers = int(sys.argv[4]) if len(sys.argv) > 4 else None

g = torch.Generator(device="cuda").manual_seed(0)
pool = 4 * batch
x = torch.randn((pool, 3, res // 16, res // 16), device="cuda", generator=g)
img = torch.nn.functional.interpolate(x, size=(res, res), mode="bicubic")
img = (img * 60 + 128 + torch.randn(img.shape, device="cuda", generator=g) * 5).clamp(0, 255).to(torch.uint8)
img = img.permute(0, 2, 3, 1).contiguous()
m = torch.nn.functional.interpolate(torch.randn((pool, 1, 8, 8), device="cuda", generator=g), size=(res, res), mode="bicubic")
mask = (m[:, 0] > 0).to(torch.uint8).contiguous()

for gpu_jpeg, gpu_png in ((False, False), (True, False), (True, True)):
    with tempfile.TemporaryDirectory() as d:
        with DatasetWriter(d, workers=workers, gpu_jpeg=gpu_jpeg, gpu_png=gpu_png) as w:   # warm-up: buffers, encoders, pool
            w.submit(img[:batch], mask[:batch], 0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        with DatasetWriter(d, workers=workers, gpu_jpeg=gpu_jpeg, gpu_png=gpu_png) as w:
            for k in range(pairs // batch):
                o = (k * batch) % pool
                w.submit(img[o:o + batch], mask[o:o + batch], k * batch)
        dt = time.perf_counter() - t
        size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / (len(os.listdir(d)) / 2.0)
        print("%d^2, batch %d, %s workers, jpeg on %s, png on %s: %d pairs in %.2f s = %.0f pairs/s (%.0f KB per pair on disk)"
              % (res, batch, workers or default_workers(), "GPU" if gpu_jpeg else "host", "GPU" if gpu_png else "host", pairs, dt,
                 pairs / dt, size / 1e3))
