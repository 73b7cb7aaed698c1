#!/usr/bin/env python3
"""Consecutive batches on ONE context/stream vs alternating between TWO contexts on two streams (the latency-bound
low-resolution chain at the start of batch k+1 then runs beside the full-chip high-resolution tail of batch k):
    python tools/pipeline_bench.py [gan=ffhq] [batch=8] [steps=40]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_segmentation_amd import _runtime, weights as W  # noqa: E402
from gan_segmentation_amd.image_generator import ImageGenerator  # noqa: E402

gan = sys.argv[1] if len(sys.argv) > 1 else "ffhq"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
mr = W.GAN_MAX_RES_LOG2[gan]
gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
gens = []
for _ in range(2):
    gens.append(ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=B))
z, noise = W.synthetic_inputs(gcfg, B)
z = torch.from_numpy(z).cuda()
noise = [torch.from_numpy(a).cuda() for a in noise]
R = 2 ** mr
outs = [(torch.empty((B, R, R, 3), dtype=torch.uint8, device="cuda"), torch.empty((B, R, R), dtype=torch.uint8, device="cuda"))
        for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(two):
    def step(k):
        s = k & 1 if two else 0
        with torch.cuda.stream(streams[s]):
            gens[s].generate_batch(z, noise, out=outs[s])
    for k in range(4):
        step(k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


a = run(False)
b = run(True)
a2 = run(False)
same = bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
print("%s batch %d: one context %.3f / %.3f ms per step (%.0f pairs/s); two alternating contexts %.3f ms (%.0f pairs/s); same output: %s"
      % (gan, B, a, a2, B / min(a, a2) * 1e3, b, B / b * 1e3, same))
