#!/usr/bin/env python3
"""Timing-only upper bounds with the debug-hook build (`make -C gan-segmentation_amd/csrc dbg`):
GSA_DBG=1 makes every conv input tile cache-resident, GSA_DBG=4 drops the conv epilogue stores,
5 = both.  Measured on MI355X (FFHQ batch 8, ms per step): 9.66 / 9.22 / 9.02 / 8.67 -- a perfect
memory system under the conv kernels would buy 10 %; the rest is MFMA + LDS/VALU structure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_segmentation_amd import _lib
_lib.HIP_LIBRARY = os.path.join(os.path.dirname(_lib.HIP_LIBRARY), "libgsa_hip_stamp.so")
from gan_segmentation_amd import weights as W
from gan_segmentation_amd.image_generator import ImageGenerator
gcfg, dcfg = W.generator_config(10), W.decoder_config(10)
gen = ImageGenerator.from_params(gcfg, W.synthetic_generator_params(gcfg), dcfg, W.synthetic_decoder_params(dcfg), gpu_ids=[0], batch_size=8)
z, noise = W.synthetic_inputs(gcfg, 8)
z = torch.from_numpy(z).cuda(); noise = [torch.from_numpy(a).cuda() for a in noise]
for _ in range(3): gen.generate_batch(z, noise)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): gen.generate_batch(z, noise)
torch.cuda.synchronize()
print("GSA_DBG", os.environ.get("GSA_DBG"), "ms/step", (time.perf_counter() - t) * 50)
