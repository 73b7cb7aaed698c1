"""Generates tests/golden/bench_outputs.json: SHA-256 of the (image, mask) bytes the CANONICAL C ORACLE
(oracle/c/gsa_oracle.c) produces for the first samples of bench.py's own inputs.

    python tests/golden/make_bench_hash.py

bench.py prints the same digest of what the HIP path produced in its timed configuration and says whether it
matches; tests/test_gpu_parity.py asserts it on the GPU, tests/test_oracle.py re-derives one entry on the CPU.
Inputs = bench.py's: synthetic weights (seeds 2 / 3), W.synthetic_inputs(gcfg, batch, seed_z=1000+rank,
seed_noise=2000+rank) with rank 0.  fp32 configurations only (bf16 mode is a stated tolerance, not bit equality).
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gan_segmentation_amd import weights as W   # noqa: E402
from oracle.binding import Oracle                # noqa: E402

CONFIGS = [("ffhq", 8, 2), ("ffhq", 4, 1), ("bedrooms", 64, 2), ("cars", 4, 1)]   # (gan, bench batch, samples hashed)
# Round 5: the FP32 tensors at full size too -- rgb, logits and the two largest features of batch 1 of
# tests/common.gan_setup (W.synthetic_inputs' default seeds), through the oracle's generator / decoder entry points: what
# gsa_generator_forward / gsa_decoder_forward (the NCHW export / import kernels at 1024^2) must reproduce bit for bit.
FP32_CONFIGS = ["ffhq", "cars"]


def digest(img, mask):
    return hashlib.sha256(img.tobytes() + mask.tobytes()).hexdigest()


def fp32_digests(gan):
    mr = W.GAN_MAX_RES_LOG2[gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
    z, noise = W.synthetic_inputs(gcfg, 1)
    o = Oracle(gcfg, gp, dcfg, dp)
    rgb, img, feats = o.generator(z, noise)
    logits, mask = o.decoder(feats)

    def h(a):
        return hashlib.sha256(a.tobytes()).hexdigest()
    return {"rgb_f32": h(rgb), "logits_f32": h(logits), "feature_last_f32": h(feats[-1]), "feature_second_last_f32": h(feats[-2]),
            "image_u8": h(img), "mask_u8": h(mask),
            "shapes": {"rgb": list(rgb.shape), "logits": list(logits.shape), "feature_last": list(feats[-1].shape),
                       "feature_second_last": list(feats[-2].shape)}}


def main():
    out = {}
    for gan in FP32_CONFIGS:
        out["%s_b1_fp32" % gan] = fp32_digests(gan)
        print(gan, "fp32", out["%s_b1_fp32" % gan], flush=True)
    for gan, batch, ns in CONFIGS:
        mr = W.GAN_MAX_RES_LOG2[gan]
        gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
        gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
        z, noise = W.synthetic_inputs(gcfg, batch, seed_z=1000, seed_noise=2000)
        img, mask = Oracle(gcfg, gp, dcfg, dp).generate(z[:ns], [a[:ns] for a in noise])
        out["%s_b%d" % (gan, batch)] = {"samples": [digest(img[i], mask[i]) for i in range(ns)],
                                        "mask_mean": [float(mask[i].mean()) for i in range(ns)]}
        print(gan, batch, out["%s_b%d" % (gan, batch)], flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_outputs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
