#!/usr/bin/env python3
"""Eager launches vs a captured hipGraph of one generate step at small batch sizes (is the step launch-bound?):
    python tools/graph_bench.py [gan=ffhq] [batches=1,2,4,8]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_segmentation_amd import weights as W  # noqa: E402
from gan_segmentation_amd.image_generator import ImageGenerator  # noqa: E402

gan = sys.argv[1] if len(sys.argv) > 1 else "ffhq"
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
mr = W.GAN_MAX_RES_LOG2[gan]
gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
for B in batches:
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=B)
    z, noise = W.synthetic_inputs(gcfg, B)
    z = torch.from_numpy(z).cuda()
    noise = [torch.from_numpy(a).cuda() for a in noise]
    R = 2 ** mr
    out = (torch.empty((B, R, R, 3), dtype=torch.uint8, device="cuda"), torch.empty((B, R, R), dtype=torch.uint8, device="cuda"))
    side = torch.cuda.Stream()

    def run(n, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    with torch.cuda.stream(side):
        eager = lambda: gen.generate_batch(z, noise, out=out)
        for _ in range(3):
            eager()
        t_eager = run(30, eager)
        ref = (out[0].clone(), out[1].clone())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            gen.generate_batch(z, noise, out=out)
        out[0].zero_(); out[1].zero_()
        g.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        t_graph = run(30, g.replay)
    print("%s batch %d: eager %.3f ms (%.0f pairs/s), graph %.3f ms (%.0f pairs/s), identical output: %s"
          % (gan, B, t_eager, B / t_eager * 1e3, t_graph, B / t_graph * 1e3, same))
