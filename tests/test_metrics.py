"""Evaluation arithmetic: the confusion-matrix bookkeeping of the product against the
literal restatement of the reference's histogram code (oracle/ref_metrics.py), and -- on the GPU -- the device
kernel and ``SegSolver.evaluate`` against both."""
import os

import numpy as np
import pytest

from gan_segmentation_amd.metrics import SegmentationMetric
from oracle import ref_metrics


def _random_case(rng, n, k, R, ignore_frac=0.2):
    logits = rng.standard_normal((n, k, R, R)).astype(np.float32) * 3
    labels = rng.integers(0, k, (n, R, R)).astype(np.int32)
    labels[rng.random((n, R, R)) < ignore_frac] = -1
    return logits, labels


def _confusion(logits, labels, k):
    pred = np.argmax(logits, 1)
    c = np.zeros((k, k), np.int64)
    ok = labels >= 0
    np.add.at(c, (labels[ok], pred[ok]), 1)
    return c


@pytest.mark.parametrize("k", [2, 3, 5])
def test_confusion_bookkeeping_equals_reference_histograms(k):
    rng = np.random.default_rng(k)
    logits, labels = _random_case(rng, 3, k, 32)
    m = SegmentationMetric(k, skip_bg=True)
    m.update_confusion(_confusion(logits[:2], labels[:2], k))       # two updates accumulate like the reference's
    m.update_confusion(_confusion(logits[2:], labels[2:], k))
    (_n1, _n2), (acc, miou) = m.get()
    acc_o, miou_o, _loss = ref_metrics.evaluate(logits, labels, k)
    assert acc == acc_o and miou == miou_o


def test_metric_edge_cases():
    m = SegmentationMetric(2)
    m.update_confusion(np.array([[5, 0], [0, 0]]))      # class 1 never labelled nor predicted: dropped like the reference does
    names, (acc, miou) = m.get()
    assert names == ["accuracy", "mean-iou"] and acc == pytest.approx(1.0) and np.isnan(miou)
    m.reset()
    m.update_confusion(np.array([[3, 1], [2, 4]]))
    _names, (acc, miou) = m.get()
    assert acc == pytest.approx(0.7) and miou == pytest.approx(4 / 7)
    logits = np.array([[[[0.0]], [[0.0]]]], np.float32)           # a tie -> first maximum -> class 0
    assert ref_metrics.weighted_softmax_ce(logits, np.array([[[1]]]))[0] == pytest.approx(np.log(2.0))
    assert ref_metrics.weighted_softmax_ce(logits, np.array([[[-1]]]))[0] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 3, 8])
def test_device_eval_kernel(k):
    import torch
    from gan_segmentation_amd._runtime import DeviceModel, current_stream_ptr
    rng = np.random.default_rng(10 + k)
    logits, labels = _random_case(rng, 3, k, 64)
    logits[0, :, 0, 0] = 1.5                                  # a tie: first maximum wins
    model = DeviceModel(0)
    dev = model.device
    lg = torch.from_numpy(logits).to(dev)
    lb = torch.from_numpy(labels.astype(np.int8)).to(dev)
    conf = torch.zeros((k, k), dtype=torch.int64, device=dev)
    lossf = torch.zeros((3,), dtype=torch.int64, device=dev)
    for _ in range(2):                                        # the buffers accumulate
        model.ctx.segmentation_eval(current_stream_ptr(dev), 3, k, 64, 64, lg.data_ptr(), lb.data_ptr(), conf.data_ptr(), lossf.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(conf.cpu().numpy(), 2 * _confusion(logits, labels, k))          # integers: exact
    loss = lossf.cpu().numpy().astype(np.float64) / 2.0 ** 32 / (64 * 64) / 2
    ref = ref_metrics.weighted_softmax_ce(logits, labels)
    assert np.abs(loss - ref).max() <= 2e-6 * max(1.0, ref.max())
    from gan_segmentation_amd import _lib
    with pytest.raises(_lib.GsaError, match="2..8 classes"):
        model.ctx.segmentation_eval(current_stream_ptr(dev), 3, 9, 64, 64, lg.data_ptr(), lb.data_ptr(), conf.data_ptr(), lossf.data_ptr())


@pytest.mark.gpu
def test_solver_evaluate_end_to_end(tmp_path, oracle_lib):
    """SegSolver.evaluate over annotator sample files == the oracle decoder + the reference's metric code."""
    from PIL import Image
    from gan_segmentation_amd import annotation_io, weights as W
    from gan_segmentation_amd.seg_solver import SegSolver
    from tests.common import reduced_setup
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(6, batch=3)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    _rgb, img, feats = o.generator(z, noise)
    R = img.shape[1]
    rng = np.random.default_rng(5)
    data = tmp_path / "data"
    masks = []
    for i in range(3):
        annotation_io.export_sample(str(data), i, img[i], [f[i] for f in feats])
        m = rng.choice(np.array([20, 128, 230], np.uint8), size=(R, R), p=[0.2, 0.4, 0.4])   # ignore / background / class 1
        Image.fromarray(m, "L").save(str(data / ("mask_%06d.png" % i)))
        masks.append(annotation_io.preprocess_mask(m))
    ckpt = tmp_path / "checkpoints"
    ckpt.mkdir()
    from gan_segmentation_amd import params as P
    P.save_params(str(ckpt / "checkpoint_last.params"), W.complete_decoder_params(dcfg, dp))
    solver = SegSolver(6, str(data), str(ckpt), gpu_ids=[0], in_channels=W.generator_channels(gcfg))
    assert solver.is_trained
    out_dir = tmp_path / "eval_out"
    result = dict(solver.evaluate(str(data), output_dir=str(out_dir)))
    # oracle side: logits of the canonical decoder on the SAME stored features (pickles hold them exactly)
    logits_o, _mask_o = o.decoder(feats)
    labels = np.stack(masks)
    acc_o, miou_o, loss_o = ref_metrics.evaluate(logits_o, labels, dcfg["num_classes"])
    assert result["accuracy"] == acc_o and result["mean-iou"] == miou_o          # integer counts behind both
    assert abs(result["total-loss"] - loss_o.mean()) <= 2e-6 * max(1.0, loss_o.mean())
    for i in range(3):
        for name in ("img_%06d.jpg", "mask_%06d.png", "gt_mask_%06d.png", "metrics_%06d.txt"):
            assert os.path.exists(str(out_dir / (name % i)))
    gt = np.asarray(Image.open(str(out_dir / "gt_mask_000001.png")))
    assert set(np.unique(gt)) <= {0, 128, 255} and np.array_equal(gt == 255, masks[1] == 1)
