"""GPU: bit-exactness of the F(4x4,3x3) layers against the C oracle (full size, batch 2)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_segmentation_amd import _lib
if os.environ.get("GSA_LIB"):
    _lib.HIP_LIBRARY = os.path.join(os.path.dirname(_lib.HIP_LIBRARY), os.environ["GSA_LIB"])
from tests.common import gan_setup
from gan_segmentation_amd.image_generator import ImageGenerator
from oracle.binding import Oracle
gan = sys.argv[1] if len(sys.argv) > 1 else "bedrooms"
gcfg, gp, dcfg, dp, z, noise = gan_setup(gan, 2)
gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=2)
rgb, feats, img = gen.netG(z, noise=noise, want_image=True)
logits, mask = gen._decoder(*feats, want_mask=True)
torch.cuda.synchronize()
o = Oracle(gcfg, gp, dcfg, dp)
orgb, oimg, ofeats = o.generator(z, noise)
ologits, omask = o.decoder(ofeats)
for i, (f, of) in enumerate(zip(feats, ofeats)):
    f = f.cpu().numpy()
    print("feat %d %s equal=%s maxdiff=%.3g" % (i, f.shape, np.array_equal(f, of), np.abs(f - of).max()), flush=True)
print("rgb equal", np.array_equal(rgb.cpu().numpy(), orgb), "logits equal", np.array_equal(logits.cpu().numpy(), ologits),
      "maxdiff", np.abs(logits.cpu().numpy() - ologits).max(), "mask equal", np.array_equal(mask.cpu().numpy(), omask))
