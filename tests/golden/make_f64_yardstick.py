"""Generates tests/golden/f64_yardstick.npz: the reference-order restatement (oracle/ref_semantic.py) evaluated in FLOAT64 for
bedrooms 256^2, cars 512^2 and ffhq 1024^2 (the headline configuration; added in round 4) at full size (tests.common.gan_setup
inputs, batch 1), stored as strided samples of rgb and logits (< 1 MB), together with the error of the SAME restatement in fp32
at those points.

    python tests/golden/make_f64_yardstick.py [gan ...]     (a few minutes of CPU; named GANs are recomputed and merged
                                                             into the existing file, no argument = all three)

It gives the north-star tolerance ("<= 1e-3 vs the reference mxnet CPU path") a yardstick: the fp32 reference order is itself
max|sem32 - f64| away from the exact result; tests/test_gpu_parity.py::test_accuracy_against_the_fp64_yardstick asserts that the
HIP path (Winograd F(2x2,3x3) / F(2x2,2x2) forms, sub-pixel form, K split, folded AdaIN) is no further from the float64 result than
twice that -- i.e. as accurate as the reference's own evaluation order, not merely within 1e-3 of it."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_semantic as S      # noqa: E402
from tests.common import gan_setup         # noqa: E402

STRIDE = {"bedrooms": 4, "cars": 8, "ffhq": 16}


def main():
    path = os.path.join(ROOT, "tests", "golden", "f64_yardstick.npz")
    out = {}
    which = sys.argv[1:] or list(STRIDE)
    if os.path.exists(path) and sys.argv[1:]:
        with np.load(path) as g:
            out = {k: g[k] for k in g.files}
    for gan in which:
        st = STRIDE[gan]
        gcfg, gp, dcfg, dp, z, noise = gan_setup(gan, 1)
        _i, _m, rgb64, _f, log64 = S.generate(gcfg, gp, dcfg, dp, z, noise, dtype=torch.float64)
        _i, _m, rgb32, _f, log32 = S.generate(gcfg, gp, dcfg, dp, z, noise)
        r64, l64 = np.asarray(rgb64, np.float64)[:, :, ::st, ::st], np.asarray(log64, np.float64)[:, :, ::st, ::st]
        r32, l32 = np.asarray(rgb32, np.float64)[:, :, ::st, ::st], np.asarray(log32, np.float64)[:, :, ::st, ::st]
        out[gan + "_rgb"] = r64
        out[gan + "_logits"] = l64
        out[gan + "_stride"] = np.int64(st)
        out[gan + "_sem32_err_rgb"] = np.float64(np.abs(r32 - r64).max())
        out[gan + "_sem32_err_logits"] = np.float64(np.abs(l32 - l64).max())
        print(gan, "stride", st, "rgb range", float(np.abs(r64).max()), "fp32 reference-order error: rgb %.3g logits %.3g"
              % (out[gan + "_sem32_err_rgb"], out[gan + "_sem32_err_logits"]), flush=True)
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
