#!/usr/bin/env python3
"""Measurement of the decoder training step: ms per optimisation step at the
reference's setting (batch 1, FFHQ 1024^2 features by default) and the per-operator split by HIP events.
    python tools/train_bench.py [ffhq|cars|bedrooms] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gan_segmentation_amd import weights as W
from gan_segmentation_amd import train_ops as ops
from gan_segmentation_amd.trainer import DecoderTrainer

gan = sys.argv[1] if len(sys.argv) > 1 else "ffhq"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mr = W.GAN_MAX_RES_LOG2[gan]
dcfg = W.decoder_config(mr)
chans = dcfg["in_channels"]
rng = np.random.default_rng(0)
feats = [torch.from_numpy(rng.standard_normal((1, c, 4 << i, 4 << i)).astype(np.float32)).cuda() for i, c in enumerate(chans)]
R = 4 << (len(chans) - 1)
labels = torch.from_numpy(rng.integers(-1, 2, (1, R, R)).astype(np.int8)).cuda()
tr = DecoderTrainer(dcfg, W.initial_decoder_params(dcfg, 1), lr=1e-4, seed=1)
tr.step(feats, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = [tr.step(feats, labels) for _ in range(steps)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# reference FLOPs of the decoder: forward 2*MACs per sample; backward = dgrad (not for the detached cvt inputs) + wgrad
fwd = {"ffhq": 62.56, "cars": 30.82, "bedrooms": 9.97}[gan]
print({"gan": gan, "ms_per_step": round(dt * 1e3, 2), "steps": steps, "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4),
       "forward_gflop_per_sample": fwd, "approx_tflops": round(3 * fwd / dt / 1e3, 2)})
