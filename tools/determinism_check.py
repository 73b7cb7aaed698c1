"""Repeat gsa_generate on the same inputs and compare every result with the first one (byte for byte): a cheap stress of the
tagged-word exchange of the mapping kernel and of the atomic statistics rows.  Usage: python tools/determinism_check.py [iters]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gan_segmentation_amd import weights as W  # noqa: E402
from gan_segmentation_amd.image_generator import ImageGenerator  # noqa: E402


def run(gan, batch, iters, precision="fp32"):
    mr = W.GAN_MAX_RES_LOG2[gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gen = ImageGenerator.from_params(gcfg, W.synthetic_generator_params(gcfg, seed=2), dcfg, W.synthetic_decoder_params(dcfg, seed=3),
                                     gpu_ids=[0], batch_size=batch, precision=precision)
    z, noise = W.synthetic_inputs(gcfg, batch, seed_z=1, seed_noise=2)
    z = torch.from_numpy(z).cuda()
    noise = [torch.from_numpy(a).cuda() for a in noise]
    first, bad = None, 0
    for i in range(iters):
        img, mask = gen.generate_batch(z, noise)
        torch.cuda.synchronize()
        h = hashlib.sha256(img.cpu().numpy().tobytes() + mask.cpu().numpy().tobytes()).hexdigest()
        if first is None:
            first = h
        elif h != first:
            bad += 1
    print("%s batch %d %s: %d iterations, %d differ from the first" % (gan, batch, precision, iters, bad), flush=True)
    return bad


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    total = run("ffhq", 8, iters) + run("ffhq", 1, 3 * iters) + run("bedrooms", 64, max(10, iters // 4)) + run("cars", 4, iters, "bf16")
    sys.exit(1 if total else 0)
