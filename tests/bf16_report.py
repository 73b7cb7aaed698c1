#!/usr/bin/env python3
"""Diagnostic script (test infrastructure: it drives the oracle; run as `python tests/bf16_report.py [gan] [batch]`):
bf16-MFMA mode of the HIP path against the C oracle in bf16 mode and against the fp32 path
(reduced 128^2 model by default, `cars`/`ffhq`/`bedrooms` for the full sizes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.common import reduced_setup, gan_setup
from oracle import binding
from gan_segmentation_amd.image_generator import ImageGenerator

which = sys.argv[1] if len(sys.argv) > 1 else "reduced"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=batch) if which == "reduced" else gan_setup(which, batch)
res = {}
for prec in ("fp32", "bf16"):
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=batch, precision=prec)
    rgb, feats, img = gen.netG(z, noise=noise, want_image=True)
    logits, mask = gen._decoder(*feats, want_mask=True)
    img2, mask2 = gen.generate_batch(z, noise)
    assert torch.equal(img, img2) and torch.equal(mask, mask2), "fused != two-call in %s" % prec
    res[prec] = [rgb.cpu().numpy(), logits.cpu().numpy(), mask.cpu().numpy(), [f.cpu().numpy() for f in feats]]
t = time.time()
o = binding.Oracle(gcfg, gp, dcfg, dp, precision="bf16")
rgb_o, img_o, feats_o = o.generator(z[:1], [a[:1] for a in noise])
logits_o, mask_o = o.decoder(feats_o)
print("oracle bf16 sample 0: %.1f s" % (time.time() - t))
def rep(tag, a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    print("%-28s max|d| %.3e  mean|d| %.3e  (max|ref| %.3e)" % (tag, d.max(), d.mean(), np.abs(b).max()))
rgb, logits, mask, feats = res["bf16"]
rep("rgb  gpu-bf16 vs oracle-bf16", rgb[:1], rgb_o)
rep("logit gpu-bf16 vs oracle-bf16", logits[:1], logits_o)
for i, (a, b) in enumerate(zip(feats, feats_o)):
    rep("feat%d gpu-bf16 vs oracle-bf16" % i, a[:1], b)
print("mask mismatch vs oracle-bf16: %.3e" % np.mean(mask[:1] != mask_o))
rep("rgb  gpu-bf16 vs gpu-fp32", rgb, res["fp32"][0])
rep("logit gpu-bf16 vs gpu-fp32", logits, res["fp32"][1])
print("mask mismatch bf16 vs fp32: %.3e" % np.mean(mask != res["fp32"][2]))
