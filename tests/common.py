"""Shared builders for the parity tests."""
import numpy as np

from gan_segmentation_amd import weights as W


def reduced_setup(max_res_log2=7, batch=2, trivial_norm=False, seed=2):
    gcfg = W.reduced_generator_config(max_res_log2)
    gp = W.synthetic_generator_params(gcfg, seed=seed, trivial_norm=trivial_norm)
    dcfg = W.decoder_config(max_res_log2, in_channels=W.generator_channels(gcfg))
    dp = W.synthetic_decoder_params(dcfg, seed=seed + 1)
    z, noise = W.synthetic_inputs(gcfg, batch)
    return gcfg, gp, dcfg, dp, z, noise


def gan_setup(gan="ffhq", batch=1):
    mr = W.GAN_MAX_RES_LOG2[gan]
    gcfg = W.generator_config(mr)
    gp = W.synthetic_generator_params(gcfg, seed=2)
    dcfg = W.decoder_config(mr)
    dp = W.synthetic_decoder_params(dcfg, seed=3)
    z, noise = W.synthetic_inputs(gcfg, batch)
    return gcfg, gp, dcfg, dp, z, noise


def bench_setup(gan="ffhq", batch=8, rank=0):
    """Exactly bench.py's weights and inputs (synthetic weights seeds 2/3, inputs seeds 1000+rank / 2000+rank)."""
    mr = W.GAN_MAX_RES_LOG2[gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
    z, noise = W.synthetic_inputs(gcfg, batch, seed_z=1000 + rank, seed_noise=2000 + rank)
    return gcfg, gp, dcfg, dp, z, noise


def golden_bench_outputs():
    """tests/golden/bench_outputs.json: SHA-256 of the C oracle's (image, mask) for the first samples of bench.py's inputs."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_outputs.json")) as f:
        return json.load(f)


def pair_digest(img, mask):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(img).tobytes() + np.ascontiguousarray(mask).tobytes()).hexdigest()
