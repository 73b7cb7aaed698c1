"""MI355X-native `generate` hot path of GAN-segmentation: StyleGAN-v1 synthesis +
segmentation decoder behind the reference's Python call surface.

Modules mirror the reference's file names: ``image_generator.ImageGenerator``,
``networks_stylegan.Generator``, ``networks_seg.Decoder``, ``seg_solver.SegSolver``.
All arithmetic runs in hand-written HIP kernels (csrc/) reached through the C ABI of
include/gsa.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
