"""One and two optimisation steps of the on-device decoder trainer against the torch-autograd restatement
(oracle/ref_train.py) with the SAME dropout masks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_training_steps_match_autograd():
    import torch
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.trainer import DecoderTrainer
    from oracle import ref_train
    mr = 6                                                    # 64x64 output, 5 levels: both shortcut kinds, concat, final conv
    gcfg = W.reduced_generator_config(mr)
    chans = W.generator_channels(gcfg)
    dcfg = W.decoder_config(mr, in_channels=chans)
    dp = W.synthetic_decoder_params(dcfg, seed=4)
    rng = np.random.default_rng(0)
    n = 2
    feats = [rng.standard_normal((n, c, 4 << i, 4 << i)).astype(np.float32) for i, c in enumerate(chans)]
    R = 4 << (len(chans) - 1)
    labels = rng.integers(0, 2, (n, R, R)).astype(np.int64)
    labels[rng.random((n, R, R)) < 0.2] = -1
    tr = DecoderTrainer(dcfg, dp, lr=1e-2, seed=5)            # a large step so that the update is well above rounding
    params = {k: v.copy() for k, v in dp.items()}
    m = {k: np.zeros_like(v) for k, v in dp.items()}
    v_ = {k: np.zeros_like(v) for k, v in dp.items()}
    for step in (1, 2):
        masks = tr.dropout_masks([(n, dcfg["features"][i], 4 << i, 4 << i) for i in range(len(chans))])
        masks_np = [mk.cpu().numpy() for mk in masks]
        loss = tr.step(feats, labels, masks=masks)
        per_sample, params, m, v_, grads = ref_train.train_step(dcfg, params, feats, labels, masks_np, step, m, v_, lr=1e-2)
        assert abs(loss - per_sample.mean()) <= 2e-5 * max(1.0, per_sample.mean())
        got = tr.state_dict()
        for k in params:
            a, b = got[k], params[k]
            if k.endswith(("running_mean", "running_var")):
                tol = 1e-4 * max(1.0, np.abs(b).max())
            elif k.endswith(".bias") and ("cvt_block" in k or "base_layers" in k):
                continue     # a bias in front of BatchNorm has an exactly-zero gradient: Adam turns its rounding noise into +-lr steps
            else:
                # Adam steps are ~lr whatever the gradient's scale, so an element whose gradient is at rounding level
                # moves by +-lr with a rounding-dependent sign: compare in the bulk, against the step size
                d = np.abs(a - b)
                lr_step = 1e-2 * step
                outliers = int(np.sum(d > 0.1 * lr_step))
                assert np.median(d) <= 0.01 * lr_step and outliers <= max(2, 0.01 * d.size), \
                    "%s after step %d: median %.3e, %d of %d elements off by > 0.1 lr" % (k, step, np.median(d), outliers, d.size)
                continue
            assert np.abs(a - b).max() <= tol, "%s after step %d: %.3e vs tol %.3e" % (k, step, np.abs(a - b).max(), tol)
    assert not np.array_equal(tr.state_dict()["cvt_block_0.0.weight"], dp["cvt_block_0.0.weight"])


@pytest.mark.parametrize("features,start_res", [(None, 0), ([32, 16, 32, 64, 2], 0), (None, 2)])
def test_gradients_match_autograd_directly(features, start_res):
    """The raw gradients (before Adam's normalisation hides their scale).  The second decoder has identity
    shortcuts over a concatenated input (2 * features[i] == features[i+1]) and a 1x1 shortcut at level 0; the
    third starts at feature 2 (cfg['start_res'], reference networks_seg.py:56): the lower levels have no blocks."""
    import torch
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.trainer import DecoderTrainer
    from oracle import ref_train
    mr = 5
    gcfg = W.reduced_generator_config(mr)
    chans = W.generator_channels(gcfg)
    dcfg = W.decoder_config(mr, in_channels=chans)
    if features is not None:
        dcfg["features"] = list(features)
    dcfg["start_res"] = start_res
    dp = W.synthetic_decoder_params(dcfg, seed=6)
    assert ("cvt_block_0.0.weight" in dp) == (start_res == 0)
    rng = np.random.default_rng(1)
    feats = [rng.standard_normal((1, c, 4 << i, 4 << i)).astype(np.float32) for i, c in enumerate(chans)]
    R = 4 << (len(chans) - 1)
    labels = rng.integers(-1, 2, (1, R, R)).astype(np.int64)
    tr = DecoderTrainer(dcfg, dp, lr=0.0, seed=2)             # lr 0: parameters stay, gradients remain readable
    masks = tr.dropout_masks([(1, dcfg["features"][i], 4 << i, 4 << i) for i in range(len(chans))])
    tr.step(feats, labels, masks=masks)
    zeros = {k: np.zeros_like(v) for k, v in dp.items()}
    _ps, _p, _m, _v, grads = ref_train.train_step(dcfg, dp, feats, labels, [mk.cpu().numpy() for mk in masks], 1, zeros, zeros, lr=0.0)
    for k, gref in grads.items():
        got = tr.g[k].cpu().numpy()
        scale = max(1e-6, np.abs(gref).max())
        assert np.abs(got - gref).max() <= 2e-3 * scale + 1e-6, "%s: %.3e of %.3e" % (k, np.abs(got - gref).max(), scale)


def test_sync_batchnorm_two_ranks_equal_the_joint_batch():
    """cfg['use_sync_bn'] (reference networks_seg.py:20-21,30-31,73-74): two ranks with one sample each, exchanging the
    BatchNorm sums and the gradients, must compute what ONE trainer computes on the two-sample batch.  The ranks are two
    threads of this process on the one GPU of the box; their all-reduce is a rendezvous that adds the two tensors."""
    import threading
    import torch
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.trainer import DecoderTrainer
    mr = 5
    gcfg = W.reduced_generator_config(mr)
    chans = W.generator_channels(gcfg)
    dcfg = W.decoder_config(mr, in_channels=chans)
    dp = W.synthetic_decoder_params(dcfg, seed=8)
    rng = np.random.default_rng(4)
    feats = [rng.standard_normal((2, c, 4 << i, 4 << i)).astype(np.float32) for i, c in enumerate(chans)]
    R = 4 << (len(chans) - 1)
    labels = rng.integers(-1, 2, (2, R, R)).astype(np.int64)
    joint = DecoderTrainer(dcfg, dp, lr=0.0, seed=2)
    masks = joint.dropout_masks([(2, dcfg["features"][i], 4 << i, 4 << i) for i in range(len(chans))])
    loss_joint = joint.step(feats, labels, masks=masks)

    # world 1 with the switch on: the split calls must reproduce the fused ones
    alone = DecoderTrainer(dict(dcfg, use_sync_bn=True), dp, lr=0.0, seed=2)
    assert alone.sync_bn
    loss_alone = alone.step(feats, labels, masks=masks)
    assert abs(loss_alone - loss_joint) <= 1e-6 * max(1.0, abs(loss_joint))
    def noise_only(k):      # a bias in front of BatchNorm has an exactly-zero gradient: what is left is rounding noise
        return k.endswith(".bias") and ("cvt_block" in k or "base_layers" in k)

    for k in joint.g:
        if noise_only(k):
            continue
        a, b = alone.g[k].cpu().numpy(), joint.g[k].cpu().numpy()       # float atomics in the weight gradients: order noise only
        assert np.abs(a - b).max() <= 1e-4 * max(1e-6, np.abs(b).max()), k

    slots, barrier = [None, None], threading.Barrier(2)

    def make_all_reduce(rank):
        def all_reduce(t):
            slots[rank] = t
            barrier.wait()
            total = slots[0] + slots[1]
            torch.cuda.synchronize()
            barrier.wait()
            t.copy_(total)
        return all_reduce

    ranks = [DecoderTrainer(dcfg, dp, lr=0.0, seed=2, sync_bn=True, all_reduce=make_all_reduce(r), world=2) for r in range(2)]
    losses, errors = [None, None], []

    def run(r):
        try:
            losses[r] = ranks[r].step([f[r:r + 1] for f in feats], labels[r:r + 1], masks=[mk[r:r + 1].contiguous() for mk in masks])
        except Exception as e:      # a failing rank must not leave the other one at the barrier
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    assert abs(0.5 * (losses[0] + losses[1]) - loss_joint) <= 1e-5 * max(1.0, abs(loss_joint))
    for k in joint.g:
        if noise_only(k):
            continue
        ref = joint.g[k].cpu().numpy()
        scale = max(1e-6, np.abs(ref).max())
        for r in range(2):
            got = ranks[r].g[k].cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-4 * scale + 1e-7, "%s rank %d: %.3e of %.3e" % (k, r, np.abs(got - ref).max(), scale)
    for k in ("cvt_block_1.1.running_mean", "main_block_2.1.base_layers.4.running_var"):
        if k in joint.p:
            for r in range(2):
                assert np.allclose(ranks[r].p[k].cpu().numpy(), joint.p[k].cpu().numpy(), rtol=1e-5, atol=1e-6), k
    # per-rank statistics (the reference's default) are NOT the joint batch's: the switch matters
    plain = DecoderTrainer(dcfg, dp, lr=0.0, seed=2)
    plain.step([f[0:1] for f in feats], labels[0:1], masks=[mk[0:1].contiguous() for mk in masks])
    k = "cvt_block_1.1.running_mean"
    assert not np.allclose(plain.p[k].cpu().numpy(), joint.p[k].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_solver_fit_learns_and_saves(tmp_path):
    """SegSolver.fit over annotator sample files: the loss falls, the checkpoint reloads, evaluate improves."""
    from PIL import Image
    from gan_segmentation_amd import annotation_io, weights as W
    from gan_segmentation_amd.seg_solver import SegSolver
    mr = 6
    gcfg = W.reduced_generator_config(mr)
    chans = W.generator_channels(gcfg)
    rng = np.random.default_rng(3)
    R = 4 << (len(chans) - 1)
    data = tmp_path / "data"
    yy, xx = np.mgrid[0:R, 0:R]
    for i in range(4):
        feats = [rng.standard_normal((c, 4 << l, 4 << l)).astype(np.float32) for l, c in enumerate(chans)]
        inside = (yy - R / 2 - 3 * i) ** 2 + (xx - R / 2 + 2 * i) ** 2 < (R / 3.5) ** 2        # a disc to segment ...
        feats[-1][0] = np.where(inside, 1.5, -1.5) + 0.3 * feats[-1][0]                          # ... visible in the finest feature
        img = np.zeros((R, R, 3), np.uint8)
        annotation_io.export_sample(str(data), i, img, feats)
        m = np.where(inside, 230, 128).astype(np.uint8)
        m[:2] = 20                                                                               # a strip of ignored pixels
        Image.fromarray(m, "L").save(str(data / ("mask_%06d.png" % i)))
    ckpt = tmp_path / "checkpoints"
    solver = SegSolver(mr, str(data), str(ckpt), gpu_ids=[0], keep_weights=False, in_channels=chans)
    assert not solver.is_trained
    lines = []
    history = solver.fit(epochs=12, log=lines.append)
    assert len(history) == 12 and len(lines) == 12
    assert history[-1] < 0.5 * history[0], history
    assert solver.is_trained and (ckpt / "checkpoint_last.params").exists()
    result = dict(solver.evaluate(str(data)))
    assert result["accuracy"] > 0.9 and result["mean-iou"] > 0.75, result
    again = SegSolver(mr, str(data), str(ckpt), gpu_ids=[0], in_channels=chans)                 # the saved checkpoint loads
    assert again.is_trained and dict(again.evaluate(str(data)))["accuracy"] == result["accuracy"]


def test_cli_train_and_evaluate_at_bedrooms_size(tmp_path, capsys):
    """`main.py train` / `main.py evaluate` (reference main.py:54-73) on a BASELINE-size decoder (bedrooms, 256^2)."""
    import yaml
    from PIL import Image
    from gan_segmentation_amd import annotation_io, main as cli, weights as W
    dcfg = W.decoder_config(W.GAN_MAX_RES_LOG2["bedrooms"])
    chans = dcfg["in_channels"]
    R = 4 << (len(chans) - 1)
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:R, 0:R]
    for sub in ("data", "eval"):
        for i in range(2):
            feats = [rng.standard_normal((c, 4 << l, 4 << l)).astype(np.float32) for l, c in enumerate(chans)]
            inside = (yy - R / 2) ** 2 + (xx - R / 2 - 10 * i) ** 2 < (R / 3) ** 2
            feats[-1][:4] += np.where(inside, 2.0, -2.0)[None]
            annotation_io.export_sample(str(tmp_path / sub), i, np.zeros((R, R, 3), np.uint8), feats)
            Image.fromarray(np.where(inside, 230, 128).astype(np.uint8), "L").save(str(tmp_path / sub / ("mask_%06d.png" % i)))
    cfg = {"BASE_DIR": str(tmp_path), "GAN": "bedrooms", "GAN_DIR": str(tmp_path), "GAN_GPU_IDS": [0], "SOLVER_GPU_IDS": [0],
           "GAN_BATCH_SIZE_PER_GPU": 2, "ANNOTATION": "segmentation"}
    (tmp_path / "config.yml").write_text(yaml.safe_dump(cfg))
    assert cli.main(["evaluate", "--config", str(tmp_path / "config.yml")]) == -1          # no checkpoint yet
    assert cli.main(["train", "--config", str(tmp_path / "config.yml"), "--epochs", "6"]) == 0
    assert (tmp_path / "checkpoints" / "checkpoint_last.params").exists()
    assert cli.main(["evaluate", "--config", str(tmp_path / "config.yml")]) == 0
    out = capsys.readouterr().out
    assert "train Decoder first!" in out and "Epoch[6] Train-total-loss=" in out
    line = [l for l in out.splitlines() if l.startswith("accuracy:")][-1]
    acc = float(line.split(",")[0].split(":")[1])
    losses = [float(l.split("=")[1]) for l in out.splitlines() if "Train-total-loss=" in l]
    assert len(losses) == 6 and losses[-1] < 0.9 * losses[0], losses          # 12 Adam steps at lr 1e-4: falling, not converged
    assert acc > 0.6, line
