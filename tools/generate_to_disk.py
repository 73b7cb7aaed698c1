#!/usr/bin/env python3
"""End-to-end `main.py generate` throughput to disk with synthetic .params files:
    python tools/generate_to_disk.py [gan=ffhq] [n=128] [batch=8] [workers] [jpeg=gpu|cpu]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_segmentation_amd import main as cli
from gan_segmentation_amd import params as P
from gan_segmentation_amd import weights as W

gan = sys.argv[1] if len(sys.argv) > 1 else "ffhq"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
workers = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None
jpeg_on_gpu = (sys.argv[5] if len(sys.argv) > 5 else "gpu") == "gpu"
mr = W.GAN_MAX_RES_LOG2[gan]
with tempfile.TemporaryDirectory() as d:
    os.makedirs(os.path.join(d, "models")); os.makedirs(os.path.join(d, "exp", "checkpoints"))
    P.save_params(os.path.join(d, "models", "stylegan-%s.params" % gan), W.synthetic_generator_params(W.generator_config(mr)))
    P.save_params(os.path.join(d, "exp", "checkpoints", "checkpoint_last.params"), W.synthetic_decoder_params(W.decoder_config(mr)))
    cfg = {"BASE_DIR": os.path.join(d, "exp"), "GAN": gan, "GAN_DIR": os.path.join(d, "models"), "GAN_GPU_IDS": [0],
           "GAN_BATCH_SIZE_PER_GPU": batch, "SOLVER_GPU_IDS": [0], "ANNOTATION": "segmentation", "GENERATE_NUM": n, "JPEG_ON_GPU": jpeg_on_gpu}
    cli.generate(cfg, limit=batch, workers=workers)          # warm-up (weights, workspace, first batch)
    t = time.perf_counter()
    cli.generate(cfg, limit=n, workers=workers)
    dt = time.perf_counter() - t
    t = time.perf_counter()
    cli.generate(cfg, limit=2 * n, workers=workers)       # twice the samples: the difference is free of the model load
    dt2 = time.perf_counter() - t
    files = os.listdir(os.path.join(d, "exp", "dataset", "train_generated"))
    print("%s: %d pairs in %.2f s = %.1f pairs/s to disk (%d files, workers=%s, jpeg on %s; includes model load)" % (
        gan, n, dt, n / dt, len(files), workers, "gpu" if jpeg_on_gpu else "cpu"))
    print("%s: steady state %.1f pairs/s to disk (%d more pairs in %.2f s more)" % (gan, n / (dt2 - dt), n, dt2 - dt))
