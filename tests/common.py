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
