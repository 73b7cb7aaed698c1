"""Inference subset of ``SegSolver`` (reference seg_solver.py:16-49, 83-132, 307-349).

    solver = SegSolver(max_res_log2, path_to_data, checkpoints_dir, gpu_ids, keep_weights=False)
    if solver.is_trained: mask = solver.predict(features)    # (N,H,W,1) float32 in {0..K-1}

Training (``fit``), evaluation and the few-shot dataset are outside the `generate` hot path
 and raise NotImplementedError.
"""
import os

import numpy as np
import torch

from . import weights as _weights
from .networks_seg import Decoder


class SegSolver:
    def __init__(self, max_res_log2, path_to_data, checkpoints_dir, gpu_ids, keep_weights=True, in_channels=None,
                 precision="fp32"):
        self.path_to_data = path_to_data
        self.checkpoints_dir = checkpoints_dir
        self.keep_weights = keep_weights
        gpu_ids = list(gpu_ids)
        if len(gpu_ids) == 0:
            raise RuntimeError("the MI355X path has no CPU context: pass a gpu id "
                               "(the reference falls back to mx.cpu(), seg_solver.py:24-26)")
        if len(gpu_ids) > 1:
            raise RuntimeError("one process drives one GPU; shard across ranks with gan_segmentation_amd.dist")
        self.ctx = gpu_ids
        self.precision = precision
        self.is_trained = False
        self.params_file = None
        self.cfg = self.get_config(max_res_log2=max_res_log2, in_channels=in_channels)
        self.net = self.init_net()
        self.is_trained = self.load()

    def get_config(self, max_res_log2=9, in_channels=None):
        return _weights.decoder_config(max_res_log2, in_channels=in_channels)  # reference :83-132

    def init_net(self):
        # the reference also Xavier-initialises here (:38-46); without a checkpoint this solver
        # is simply "not trained" and predict() refuses to run
        return Decoder(self.cfg, num_devices=len(self.ctx), device=self.ctx[0], precision=self.precision)

    def load(self):
        """Load the first ``*.params`` file of the checkpoints dir (reference :339-349)."""
        if not self.checkpoints_dir or not os.path.isdir(self.checkpoints_dir):
            return False
        files = sorted(f for f in os.listdir(self.checkpoints_dir) if os.path.splitext(f)[1] == ".params")
        if not files:
            return False
        self.params_file = files[0]
        self.net.load_parameters(os.path.join(self.checkpoints_dir, files[0]))
        return True

    def load_parameters(self, source):
        self.net.load_parameters(source)
        self.is_trained = True

    def save(self, suffix=None):
        name = "checkpoint_last.params" if suffix is None else "checkpoint_%s.params" % suffix
        self.params_file = name
        os.makedirs(self.checkpoints_dir, exist_ok=True)
        self.net.save_parameters(os.path.join(self.checkpoints_dir, name))

    def predict(self, features, keep_on_device=False):
        """features: list of (C,H,W) or (N,C,H,W) arrays -> (N,H,W,1) float32 class indices
        (argmax over classes, first maximum; reference :307-329)."""
        if not self.is_trained:
            raise RuntimeError("train Decoder first! (no checkpoint loaded)")
        _logits, mask = self.net(*features, want_mask=True)
        mask = mask.unsqueeze(-1)
        if keep_on_device:
            return mask.to(torch.float32)
        torch.cuda.synchronize()
        return mask.cpu().numpy().astype(np.float32)

    def fit(self, *args, **kwargs):
        raise NotImplementedError("decoder training is outside the generate hot path (SURVEY.md section 8f)")

    def evaluate(self, *args, **kwargs):
        raise NotImplementedError("evaluation is outside the generate hot path (SURVEY.md section 8f)")
