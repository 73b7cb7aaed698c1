"""``python -m gan_segmentation_amd.main [generate|train|evaluate]`` -- the actions of reference main.py:44-104
with the same ``config.yml`` keys (reference config.yml.example:1-8); `generate` is the hot path, `train` and
`evaluate` drive ``SegSolver.fit`` / ``SegSolver.evaluate``.  The Tk `annotation` GUI
is out of scope.

Writes ``img_%06d.jpg`` (RGB image; the reference flips to BGR only because cv2 expects it)
and ``mask_%06d.png`` (single channel, class index) into BASE_DIR/dataset/train_generated.
With torchrun (one process per GPU) the sample indices are sharded across ranks and every rank
writes its own files -- no collective is needed when the sink is the filesystem.
"""
import argparse
import os
import sys

import numpy as np
import yaml


def load_config_file(path):
    with open(path) as f:
        return yaml.safe_load(f)   # reference utils.py:112-115


def devices_for_rank(cfg, world, local_rank):
    """(GAN device list, solver device list) of this process.  One process (the reference's way, main.py:79-88):
    the whole ``GAN_GPU_IDS`` / ``SOLVER_GPU_IDS`` lists, batch split over them in-process.  Under torchrun (one
    process per GPU): rank r drives ``GAN_GPU_IDS[r % len]`` alone (an empty list = device ``local_rank``)."""
    gan_ids = list(cfg.get("GAN_GPU_IDS") or [])
    solver_ids = list(cfg.get("SOLVER_GPU_IDS") or gan_ids)
    if world > 1:
        gpu = gan_ids[local_rank % len(gan_ids)] if gan_ids else local_rank
        return [gpu], [gpu]
    if not gan_ids:
        raise RuntimeError("GAN_GPU_IDS is empty: the MI355X path has no CPU context (reference image_generator.py:17)")
    return gan_ids, solver_ids[:1] if solver_ids else gan_ids[:1]


def shard_batches(n_generate, batch, world, rank):
    """(first global sample index, samples) of every batch THIS rank produces: rank r owns the contiguous slice
    ``dist.shard_bounds(n_generate, world, r)`` of the file indices and walks it in batches, the last one short
    (reference main.py:93-99 with image_generator.py:88-92).  The slices of all ranks partition [0, n_generate)."""
    from .dist import shard_bounds
    lo, hi = shard_bounds(n_generate, world, rank)
    index = lo
    while index < hi:
        bs = min(batch, hi - index)
        yield index, bs
        index += bs


def generate(cfg, limit=None, workers=None):
    import torch
    from . import dist as gdist
    from .dataset_writer import DatasetWriter, DeviceCheckFailed
    from .image_generator import ImageGenerator
    from .seg_solver import SegSolver
    from .weights import GAN_MAX_RES_LOG2

    # ranks never exchange anything here (every rank writes its own files), so no process group is created:
    # RANK / WORLD_SIZE / LOCAL_RANK alone decide the shard
    rank, world, local_rank = gdist.env_ranks()
    root_dir, gan, gan_dir = cfg["BASE_DIR"], cfg["GAN"], cfg["GAN_DIR"]
    gan_ids, solver_ids = devices_for_rank(cfg, world, local_rank)
    n_generate = cfg.get("GENERATE_NUM", 10000) if limit is None else limit
    batch = cfg["GAN_BATCH_SIZE_PER_GPU"] * len(gan_ids)      # reference main.py:87
    seed = int(cfg.get("SEED", 0))             # additive key: seed of the counter-based latents/noise
    precision = cfg.get("PRECISION", "fp32")   # additive key: "bf16" = bf16 MFMA operands (BASELINE config 5)

    solver = SegSolver(GAN_MAX_RES_LOG2[gan], os.path.join(root_dir, "data"), os.path.join(root_dir, "checkpoints"),
                       gpu_ids=solver_ids, keep_weights=False, precision=precision)
    if not solver.is_trained:
        print("train Decoder first!")   # reference main.py:82-84
        return -1
    netG = ImageGenerator(gpu_ids=gan_ids, gan_dir=gan_dir, gan=gan, batch_size=batch, precision=precision)
    netG.attach_decoder(solver.cfg, solver.net)
    dst_dir = os.path.join(root_dir, "dataset", "train_generated")
    os.makedirs(dst_dir, exist_ok=True)

    # the writer copies each batch out through pinned buffers and encodes it on a thread pool while the
    # GPU already computes the next one
    # additive keys JPEG_ON_GPU / PNG_ON_GPU (default on): the files are compressed by the HIP kernels behind
    # include/gsa_jpeg.h and include/gsa_png.h; the host threads only frame and write them
    on_gpu = bool(cfg.get("JPEG_ON_GPU", True))
    try:
        with DatasetWriter(dst_dir, workers=workers, gpu_jpeg=on_gpu, gpu_png=bool(cfg.get("PNG_ON_GPU", on_gpu))) as writer:
            for index, bs in shard_batches(n_generate, batch, world, rank):
                # latents and noise keyed on the global sample index: the files are the same for any number of ranks
                img, mask = netG.generate_indexed(index, bs, seed=seed)
                # the device-side checks travel with the batch: 8 bytes copied behind its kernels, read by the writer before it
                # releases the batch's files -- a run of 10 000 samples stops at the first bad batch instead of reporting at close()
                writer.submit(img, mask, index, status=netG.snapshot_status())
    except DeviceCheckFailed as e:
        print("generate: %s" % e, file=sys.stderr)
        return -2
    torch.cuda.synchronize()
    return 0


def export_for_annotation(cfg, count):
    """Headless stand-in for the sampling half of the annotator (reference seg_annotator.py:286-337): generate `count`
    samples and save what the GUI saves next to a drawn mask -- BASE_DIR/data/img_%06d.jpg and feat_%06d.pickle."""
    from . import annotation_io
    from .image_generator import ImageGenerator
    gpu = list(cfg["GAN_GPU_IDS"])[0]
    netG = ImageGenerator(gpu_ids=[gpu], gan_dir=cfg["GAN_DIR"], gan=cfg["GAN"], batch_size=cfg["GAN_BATCH_SIZE_PER_GPU"],
                          precision=cfg.get("PRECISION", "fp32"), seed=int(cfg.get("SEED", 0)))
    dst = os.path.join(cfg["BASE_DIR"], "data")
    for i, (img, feats) in enumerate(netG.get_images(count)):
        annotation_io.export_sample(dst, i, img, feats)
    print("wrote %d samples (img + feat) to %s" % (count, dst))
    return 0


def _solver(cfg, keep_weights=False):
    from .seg_solver import SegSolver
    from .weights import GAN_MAX_RES_LOG2
    root_dir, gan = cfg["BASE_DIR"], cfg["GAN"]
    gpu_ids = list(cfg.get("SOLVER_GPU_IDS", cfg.get("GAN_GPU_IDS", [0])))[:1]
    return SegSolver(GAN_MAX_RES_LOG2[gan], os.path.join(root_dir, "data"), os.path.join(root_dir, "checkpoints"),
                     gpu_ids=gpu_ids, keep_weights=keep_weights, precision=cfg.get("PRECISION", "fp32"))


def train(cfg, epochs=None):
    """The `train` action (reference main.py:54-60): fit the decoder on BASE_DIR/data, save the checkpoint."""
    solver = _solver(cfg, keep_weights=False)
    history = solver.fit(epochs=epochs, log=print)
    print("final loss: %.6f" % history[-1])
    return 0


def evaluate(cfg):
    """The `evaluate` action (reference main.py:61-73) over BASE_DIR/eval."""
    solver = _solver(cfg)
    if not solver.is_trained:
        print("train Decoder first!")
        return -1
    result = solver.evaluate(os.path.join(cfg["BASE_DIR"], "eval"))
    print(", ".join("%s: %.4f" % (name, value) for name, value in result))
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("action", nargs="?", choices=("annotation", "train", "evaluate", "generate"), default="generate")
    ap.add_argument("--config", default="config.yml")
    ap.add_argument("--limit", type=int, default=None, help="override GENERATE_NUM")
    ap.add_argument("--workers", type=int, default=None, help="encoder threads (default: the CPU share of the process)")
    ap.add_argument("--epochs", type=int, default=None, help="train: override the reference's 24 epochs")
    ap.add_argument("--count", type=int, default=None, help="annotation: number of samples to generate and export")
    args = ap.parse_args(argv)
    np.random.seed(0)
    cfg = load_config_file(args.config)
    if args.action == "annotation":
        if args.count is None:
            print("the Tk annotation GUI is out of scope (SURVEY.md section 8); `annotation --count N` writes the img/feat "
                  "files the GUI saves for N generated samples (masks are then drawn with any image editor)")
            return 2
        return export_for_annotation(cfg, args.count)
    if args.action == "train":
        return train(cfg, args.epochs)
    if args.action == "evaluate":
        return evaluate(cfg)
    return generate(cfg, args.limit, args.workers)


if __name__ == "__main__":
    sys.exit(main())
