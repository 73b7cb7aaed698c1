#!/usr/bin/env python3
"""Diagnostic: run one FFHQ batch through the stamp build (csrc/libgsa_hip_stamp.so, `make stamp`)
and print the per-phase wave-cycle averages of every conv launch (stderr lines `STAMP ...`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_segmentation_amd import _lib
_lib.HIP_LIBRARY = os.path.join(os.path.dirname(_lib.HIP_LIBRARY), "libgsa_hip_stamp.so")
from gan_segmentation_amd import weights as W
from gan_segmentation_amd.image_generator import ImageGenerator
gan = sys.argv[1] if len(sys.argv) > 1 else "ffhq"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
mr = W.GAN_MAX_RES_LOG2[gan]
gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
gen = ImageGenerator.from_params(gcfg, W.synthetic_generator_params(gcfg), dcfg, W.synthetic_decoder_params(dcfg), gpu_ids=[0], batch_size=B, precision=precision)
z, noise = W.synthetic_inputs(gcfg, B)
gen.netG._model.ctx.profile_enable(2)
for i in range(2):
    print("== pass", i, file=sys.stderr)
    gen.generate_batch(z, noise)
    torch.cuda.synchronize()
