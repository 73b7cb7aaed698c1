"""Regenerates tests/golden/reduced7_seed2.npz from the C oracle (run from the repo root:
`python tests/golden/make_golden.py`).  Inputs come from fixed seeds (tests/common.py);
the file stores expected outputs only (u8 image+mask in full, strided fp32 samples)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.binding import Oracle, build  # noqa: E402
from tests.common import reduced_setup  # noqa: E402

build()
gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=2, trivial_norm=False)
o = Oracle(gcfg, gp, dcfg, dp)
rgb, img, feats = o.generator(z, noise)
logits, mask = o.decoder(feats)
out = {"img": img, "mask": mask, "rgb_sub": rgb[:, :, ::8, ::8], "logits_sub": logits[:, :, ::8, ::8]}
for i, f in enumerate(feats):
    out["feat%d_corner" % i] = f[:, :4, :4, :4]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reduced7_seed2.npz"), **out)
print("wrote golden vectors:", {k: v.shape for k, v in out.items()})
