"""Inference subset of ``SegSolver`` (reference seg_solver.py:16-49, 83-132, 307-349).

    solver = SegSolver(max_res_log2, path_to_data, checkpoints_dir, gpu_ids, keep_weights=False)
    if solver.is_trained: mask = solver.predict(features)    # (N,H,W,1) float32 in {0..K-1}

``evaluate`` (pixAcc / mIoU / weighted softmax-CE over annotated samples) and ``fit``
(section 8f-3: the decoder's training loop, ``trainer.DecoderTrainer``) run on the device too.
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
        self.ctx = gpu_ids      # the reference's device list (seg_solver.py:24-28): one decoder replica per entry
        self.precision = precision
        self.is_trained = False
        self.params_file = None
        self.cfg = self.get_config(max_res_log2=max_res_log2, in_channels=in_channels)
        self.nets = [self.init_net(dev) for dev in self.ctx]
        self.net = self.nets[0]
        self.is_trained = self.load()

    def get_config(self, max_res_log2=9, in_channels=None):
        return _weights.decoder_config(max_res_log2, in_channels=in_channels)  # reference :83-132

    def init_net(self, device=None):
        # the reference also Xavier-initialises here (:38-46); without a checkpoint this solver
        # is simply "not trained" and predict() refuses to run
        return Decoder(self.cfg, num_devices=len(self.ctx), device=self.ctx[0] if device is None else device,
                       precision=self.precision)

    def load(self):
        """Load the first ``*.params`` file of the checkpoints dir (reference :339-349)."""
        if not self.checkpoints_dir or not os.path.isdir(self.checkpoints_dir):
            return False
        files = sorted(f for f in os.listdir(self.checkpoints_dir) if os.path.splitext(f)[1] == ".params")
        if not files:
            return False
        self.params_file = files[0]
        self.load_parameters(os.path.join(self.checkpoints_dir, files[0]))
        return True

    def load_parameters(self, source):
        from . import params as _params
        tensors = _params.load_params(source) if isinstance(source, (str, bytes)) else dict(source)
        for net in self.nets:       # full replica per device (reference: load_parameters(..., ctx=self.ctx))
            net.load_parameters(tensors)
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
        n = 1 if np.ndim(features[0]) == 3 else len(features[0])
        if len(self.nets) == 1 or n == 1:
            _logits, mask = self.net(*features, want_mask=True)
        else:
            # split over the device list, gather on the first device (reference :317-325)
            from ._runtime import split_sizes
            parts = []
            for r, lo, hi in split_sizes(n, len(self.nets)):
                net = self.nets[r]
                with torch.cuda.device(net._model.device):
                    parts.append(net(*[f[lo:hi] for f in features], want_mask=True)[1])
            dev0 = self.net._model.device
            mask = torch.cat([m.to(dev0) for m in parts], dim=0)
        mask = mask.unsqueeze(-1)
        if keep_on_device:
            return mask.to(torch.float32)
        for net in self.nets:
            torch.cuda.synchronize(net._model.device)
        return mask.cpu().numpy().astype(np.float32)

    # -- training ---------------------------------------------------------
    def fit(self, epoch_end_callback=None, epochs=None, seed=None, log=None):
        """Train the decoder on the annotated samples of ``path_to_data`` (``feat_*.pickle`` + ``mask_*.png``) as
        reference ``fit`` (:351-465): batch size 1, ``train_epochs`` epochs (24) of shuffled samples, Adam(1e-4),
        weighted softmax-CE, then ``save()``.  Every operator runs on the MI355X (``trainer.DecoderTrainer``).
        ``keep_weights=False`` starts from a fresh Xavier initialisation as the reference does.  Returns the mean
        loss of each epoch."""
        import random
        from . import annotation_io
        from .trainer import DecoderTrainer
        names = sorted(f for f in os.listdir(self.path_to_data) if f.endswith(".pickle") and "feat" in f)
        if not names:
            raise ValueError("number of training samples should be > 0")          # reference :139-141
        seed = 1 if seed is None else seed                                          # reference cfg['seed'] = 1
        if self.keep_weights and getattr(self.net, "_tensors", None) is not None:
            start = self.net._tensors
        else:
            start = _weights.initial_decoder_params(self.cfg, seed)
        tr = DecoderTrainer(self.cfg, start, device=self.ctx[0], lr=1e-4, wd=0.0, seed=seed)     # reference cfg: adam, base_lr 1e-4, wd 0
        epochs = 24 if epochs is None else epochs                                                # reference cfg['train_epochs']
        rng = random.Random(seed)
        history = []
        import torch.distributed as tdist
        world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
        rank = tdist.get_rank() if world > 1 else 0
        if world > 1 and len(names) < world:
            # every rank would train on the full list while all-reducing gradients: the same samples `world` times per step
            raise ValueError("data-parallel fit: %d training samples for %d ranks -- use at most one rank per sample" % (len(names), world))
        for epoch in range(epochs):
            order = list(names)
            rng.shuffle(order)           # the same permutation on every rank (same seed) ...
            if world > 1:                           # ... of which rank r takes every world-th sample: the trainer
                order = order[:len(order) - len(order) % world][rank::world]   # all-reduces the gradients, so a step sees `world` samples
            total = 0.0
            for fname in order:
                image_id = int(os.path.splitext(fname)[0].split("_")[-1])
                mask, _img, feats = annotation_io.load_sample(self.path_to_data, image_id)
                if mask is None:
                    raise ValueError("no mask for %s" % fname)
                total += tr.step(feats, mask[None])
            history.append(total / len(order))
            if log is not None:
                log("Epoch[%d] Train-total-loss=%f" % (epoch + 1, history[-1]))
            if epoch_end_callback is not None:
                epoch_end_callback()
        self.load_parameters(tr.state_dict())
        # rank 0 writes the checkpoint; the other ranks must not go on to load() it before it is complete -- and must not wait
        # for ever if the write fails (disk full, bad path): the ranks exchange a success flag in place of a plain barrier and
        # every rank raises
        save_error = None
        if rank == 0:
            try:
                self.save()
            except Exception as e:       # noqa: BLE001 -- re-raised below, on every rank
                save_error = e
        if world > 1:
            flag = torch.tensor([0 if save_error is None else 1], dtype=torch.int32,
                                device=self.net._model.device if tdist.get_backend() == "nccl" else "cpu")
            tdist.all_reduce(flag, op=tdist.ReduceOp.MAX)
            if int(flag.item()) and save_error is None:
                raise RuntimeError("fit: rank 0 could not write the checkpoint (see its error)")
        if save_error is not None:
            raise save_error
        return history

    # -- evaluation ------------------------------------------------------
    def evaluate_batch(self, features, labels, metric=None):
        """One batch of reference ``evaluate_for_data`` (:229-262) on the device: decoder forward, weighted
        softmax-CE and the confusion counts of ``SegmentationMetric.update``.  ``labels``: (N,H,W) integers,
        -1 = ignore.  Returns (batch-mean loss, confusion (K,K) int64); updates ``metric`` when given."""
        from ._runtime import current_stream_ptr
        logits, _mask = self.net(*features, want_mask=True)
        n, k, H, W = logits.shape
        dev = logits.device
        lab = torch.as_tensor(np.asarray(labels)).reshape(n, H, W).to(device=dev, dtype=torch.int8).contiguous()
        conf = torch.zeros((k, k), dtype=torch.int64, device=dev)
        loss_fixed = torch.zeros((n,), dtype=torch.int64, device=dev)
        self.net._model.ctx.segmentation_eval(current_stream_ptr(dev), n, k, H, W, logits.data_ptr(), lab.data_ptr(),
                                              conf.data_ptr(), loss_fixed.data_ptr())
        conf_np = conf.cpu().numpy()
        per_sample = loss_fixed.cpu().numpy().astype(np.float64) / 2.0 ** 32 / (H * W)
        if metric is not None:
            metric.update_confusion(conf_np)
        return float(per_sample.mean()), conf_np

    def evaluate(self, input_dir, output_dir=None):
        """pixAcc / mIoU / loss over the annotated samples of ``input_dir`` (``feat_*.pickle`` + ``img_*.jpg`` +
        ``mask_*.png``, reference seg_datasets.py) -> [('accuracy', v), ('mean-iou', v), ('total-loss', v)]
        (reference :222-305).  Samples are evaluated one per batch (the reference's loss is the mean of
        batch means, identical for full batches).  With ``output_dir`` the per-sample image, predicted mask
        (255/128), ground truth (255/128/0) and a metrics line are written as the reference does."""
        from . import annotation_io
        from .metrics import SegmentationMetric
        if not self.is_trained:
            raise RuntimeError("train Decoder first! (no checkpoint loaded)")
        names = sorted(f for f in os.listdir(input_dir) if f.endswith(".pickle") and "feat" in f)
        if not names:
            raise ValueError("number of eval samples should be > 0")     # reference :160-162
        nclass = self.cfg["num_classes"]
        metric = SegmentationMetric(nclass, skip_bg=True)
        total_loss, total_cnt = 0.0, 0
        if output_dir is not None:
            os.makedirs(output_dir, exist_ok=True)
        for fname in names:
            image_id = int(os.path.splitext(fname)[0].split("_")[-1])
            mask, img, feats = annotation_io.load_sample(input_dir, image_id)
            if mask is None:
                raise ValueError("no mask for %s" % fname)
            loss_v, conf = self.evaluate_batch(feats, mask[None])
            metric.update_confusion(conf)
            total_loss += loss_v
            total_cnt += 1
            if output_dir is not None:
                self._write_eval_sample(output_dir, image_id, img, feats, mask, conf, nclass)
        result = metric.get_name_value()
        result.append(("total-loss", total_loss / total_cnt if total_cnt else 0.0))
        return result

    def _write_eval_sample(self, output_dir, image_id, img, feats, mask, conf, nclass):
        from PIL import Image
        from .metrics import SegmentationMetric
        m = SegmentationMetric(nclass, skip_bg=True)
        m.update_confusion(conf)
        metric_str = ", ".join("%s %.3f" % (name, v) for name, v in m.get_name_value())
        pred = self.predict(feats)[0, :, :, 0].astype(np.int32)
        pred_img = np.where(pred == 1, 255, np.where(pred == 0, 128, pred)).astype(np.uint8)
        gt = np.where(mask == 1, 255, np.where(mask == 0, 128, 0)).astype(np.uint8)
        imname = "img_%06d.jpg" % image_id
        Image.fromarray(np.ascontiguousarray(img, np.uint8), "RGB").save(os.path.join(output_dir, imname))
        Image.fromarray(pred_img, "L").save(os.path.join(output_dir, "mask_%06d.png" % image_id))
        Image.fromarray(gt, "L").save(os.path.join(output_dir, "gt_mask_%06d.png" % image_id))
        with open(os.path.join(output_dir, "metrics_%06d.txt" % image_id), "w") as fp:
            fp.write(", ".join(str(w) for w in [imname, img.shape, pred_img.shape, gt.shape, metric_str]) + "\n")
