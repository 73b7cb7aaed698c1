#!/usr/bin/env python3
"""Debug aid: where does a GSA_WINO_LEAN build differ from conv3x3_wino?  Runs the reduced 128 px configuration in two child
processes (the switch is read once per process), saves rgb + features, and says per feature whether the difference is per-channel
affine (statistics) or pixel-wise (values).   usage: python tools/dbg_lean.py [lean bits, default 3]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from tests.common import reduced_setup
from gan_segmentation_amd.image_generator import ImageGenerator
gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=2, trivial_norm=False)
gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=2)
rgb, feats, img = gen.netG(z, noise=noise, want_image=True)
logits, mask = gen._decoder(*feats, want_mask=True)
np.savez(sys.argv[1], rgb=rgb.cpu().numpy(), logits=logits.cpu().numpy(), **{"f%%d" %% i: f.cpu().numpy() for i, f in enumerate(feats)})
''' % ROOT
bits = sys.argv[1] if len(sys.argv) > 1 else "3"
out = {}
for tag, v in (("ref", "0"), ("lean", bits)):
    path = "/tmp/dbg_lean_%s.npz" % tag
    subprocess.check_call([sys.executable, "-c", WORKER, path], env=dict(os.environ, GSA_WINO_LEAN=v))
    out[tag] = np.load(path)
for k in out["ref"].files:
    a, b = out["ref"][k].astype(np.float64), out["lean"][k].astype(np.float64)
    d = np.abs(a - b)
    msg = "%-7s shape %-18s max|d| %.3e" % (k, a.shape, d.max())
    if d.max() > 0 and a.ndim == 4:
        # per (sample, channel) least-squares fit b = s*a + t: residual ~ 0 means an affine (statistics) difference
        res = 0.0
        for n in range(a.shape[0]):
            for c in range(a.shape[1]):
                x, y = a[n, c].ravel(), b[n, c].ravel()
                A = np.stack([x, np.ones_like(x)], 1)
                coef, *_ = np.linalg.lstsq(A, y, rcond=None)
                res = max(res, np.abs(A @ coef - y).max())
        msg += "  residual after a per-channel affine fit %.3e" % res
        bad = np.argwhere(d > 1e-6)
        if len(bad):
            msg += "  first diff at %s" % (tuple(bad[0]),)
    print(msg, flush=True)
