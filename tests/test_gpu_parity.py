"""GPU parity tests: the HIP path (through the C ABI) against the canonical C oracle.

The bar is BIT-EXACT equality for every output (fp32 rgb, features, logits, u8 image, u8 mask):
the kernels implement the oracle's canonical fp32 evaluation order."""
import os

import numpy as np
import pytest

from tests.common import reduced_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _first_diff(a, b):
    idx = np.argwhere(a != b)
    return "%d of %d differ, first at %s: %r vs %r, max abs %g" % (
        len(idx), a.size, tuple(idx[0]), a[tuple(idx[0])], b[tuple(idx[0])],
        np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


def assert_same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    assert np.array_equal(a, b), "%s: %s" % (what, _first_diff(a, b))


def test_reduced_generator_decoder_bit_exact(torch_cuda, oracle_lib):
    from gan_segmentation_amd.networks_seg import Decoder
    from gan_segmentation_amd.networks_stylegan import Generator
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=3)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    rgb_o, img_o, feats_o = o.generator(z, noise)
    logits_o, mask_o = o.decoder(feats_o)

    g = Generator(gcfg)
    g.load_parameters(gp)
    d = Decoder(dcfg, 1)
    d.load_parameters(dp)
    rgb, feats, img = g(z, noise=noise, want_image=True)
    for i, (a, b) in enumerate(zip(feats, feats_o)):
        assert_same(a.cpu().numpy(), b, "feature %d" % i)
    assert_same(rgb.cpu().numpy(), rgb_o, "rgb")
    assert_same(img.cpu().numpy(), img_o, "image")
    logits, mask = d(*feats, want_mask=True)
    assert_same(logits.cpu().numpy(), logits_o, "logits")
    assert_same(mask.cpu().numpy(), mask_o, "mask")


def _build(gcfg, gp, dcfg, dp, batch):
    from gan_segmentation_amd.image_generator import ImageGenerator
    return ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=batch)


def test_fused_generate_equals_two_calls_and_oracle(torch_cuda, oracle_lib):
    """gsa_generate (features stay in the kernels' layout) == generator_forward+decoder_forward
    == oracle, for the non-trivial-norm weights (random IN gamma/beta, perturbed blur taps)."""
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=5, trivial_norm=False)
    gen = _build(gcfg, gp, dcfg, dp, 5)
    img, mask = gen.generate_batch(z, noise)
    rgb, feats, img2 = gen.netG(z, noise=noise, want_image=True)
    _logits, mask2 = gen._decoder(*feats, want_mask=True)
    assert_same(img.cpu().numpy(), img2.cpu().numpy(), "fused vs two-call image")
    assert_same(mask.cpu().numpy(), mask2.cpu().numpy(), "fused vs two-call mask")
    img_o, mask_o = oracle_lib.Oracle(gcfg, gp, dcfg, dp).generate(z, noise)
    assert_same(img.cpu().numpy(), img_o, "image vs oracle")
    assert_same(mask.cpu().numpy(), mask_o, "mask vs oracle")


def test_batch_composition_does_not_change_a_sample(torch_cuda):
    """Sharding property used by the multi-GPU path: a sample's result depends only on its own
    (z_i, noise_i), so any split of a batch over ranks reproduces the single-GPU bytes."""
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=6)
    gen = _build(gcfg, gp, dcfg, dp, 6)
    img, mask = gen.generate_batch(z, noise)
    img, mask = img.cpu().numpy(), mask.cpu().numpy()
    for lo, hi in ((0, 1), (1, 4), (4, 6)):
        i2, m2 = gen.generate_batch(z[lo:hi], [a[lo:hi] for a in noise])
        assert_same(i2.cpu().numpy(), img[lo:hi], "image shard %d:%d" % (lo, hi))
        assert_same(m2.cpu().numpy(), mask[lo:hi], "mask shard %d:%d" % (lo, hi))


def test_semantic_tolerance(torch_cuda):
    """North-star tolerance against the reference-order restatement (torch functionals):
    max |rgb diff| <= 1e-3 fp32; masks equal except where the two logits nearly tie."""
    from oracle import ref_semantic as S
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=2)
    gen = _build(gcfg, gp, dcfg, dp, 2)
    rgb, feats = gen.netG(z, noise=noise)
    logits, mask = gen._decoder(*feats, want_mask=True)
    simg, smask, srgb, sfeats, slog = S.generate(gcfg, gp, dcfg, dp, z, noise)
    assert np.abs(rgb.cpu().numpy() - srgb).max() <= 1e-3
    assert np.abs(logits.cpu().numpy() - slog).max() <= 1e-3
    margin = np.abs(slog[:, 1] - slog[:, 0])
    differ = mask.cpu().numpy() != smask
    assert not (differ & (margin > 1e-3)).any()


def test_reference_surface_get_images_and_predict(torch_cuda, oracle_lib, tmp_path):
    """ImageGenerator.get_images / SegSolver.predict (reference image_generator.py:86-124,
    seg_solver.py:307-329) through .params files on disk."""
    from gan_segmentation_amd import params as P
    from gan_segmentation_amd.image_generator import ImageGenerator
    from gan_segmentation_amd.seg_solver import SegSolver
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=3)
    ckpt = tmp_path / "checkpoints"
    ckpt.mkdir()
    P.save_params(str(ckpt / "checkpoint_last.params"), dp)
    gen = ImageGenerator.from_params(gcfg, gp, gpu_ids=[0], batch_size=2, return_latents=True)
    solver = SegSolver(7, str(tmp_path / "data"), str(ckpt), gpu_ids=[0], keep_weights=False,
                       in_channels=dcfg["in_channels"])
    assert solver.is_trained
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    _rgb_o, img_o, feats_o = o.generator(z, noise)
    _log_o, mask_o = o.decoder(feats_o)
    got = list(gen.get_images(3, latents=z, noise=noise))
    assert len(got) == 3
    for i, (img, feats, lat) in enumerate(got):
        assert img.shape == (128, 128, 3) and img.dtype == np.uint8
        assert_same(img, img_o[i], "get_images image %d" % i)
        assert lat.shape[0] == (2 if i < 2 else 1)        # the WHOLE batch array, reference :105,122
        for f, fo in zip(feats, feats_o):
            assert_same(f, fo[i], "get_images feature")
        m = solver.predict(feats)                            # 3-D (single sample) features
        assert m.shape == (1, 128, 128, 1) and m.dtype == np.float32
        assert_same(m[0, :, :, 0].astype(np.uint8), mask_o[i], "predict mask %d" % i)


def test_error_paths(torch_cuda):
    from gan_segmentation_amd import _lib
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.networks_stylegan import Generator
    gcfg = W.reduced_generator_config(7)
    gp = W.synthetic_generator_params(gcfg)
    g = Generator(gcfg)
    with pytest.raises(RuntimeError):
        g(np.zeros((1, 512), np.float32))                    # parameters not loaded
    bad = dict(gp)
    del bad["64_conv_2_weight"]
    with pytest.raises(KeyError):
        g.load_parameters(bad)                               # no allow_missing in the reference
    g.load_parameters(dict(gp, extra_key=np.zeros(3, np.float32)))   # ignore_extra=True
    with pytest.raises(ValueError):
        g(np.zeros((1, 100), np.float32))
    ctx = _lib.Context(_lib.load_library(), 0)
    with pytest.raises(_lib.GsaError):
        ctx.generator_init(W.generator_config(7, fmap_base=64, fmap_max=8))   # channels not multiple of 16


@pytest.mark.parametrize("gan,batch", [("bedrooms", 3), ("cars", 1), ("ffhq", 1)])
def test_full_size_bit_exact(torch_cuda, oracle_lib, gan, batch):
    """BASELINE.json full-size configurations (synthetic weights) against the C oracle: the u8 pairs of the fused path AND
    (round 5) every FP32 tensor of the two-call path -- rgb, all features, logits through
    gsa_generator_forward / gsa_decoder_forward, i.e. through the NCHW export / import kernels at full size (32 % of the u8
    image values of the synthetic-weights ffhq sample are saturated: the u8 projection alone could hide an error below 1e-3).
    Batch 1 of ffhq / cars also equals the committed digests of tests/golden/bench_outputs.json (make_bench_hash.py)."""
    import hashlib
    from tests.common import gan_setup, golden_bench_outputs
    gcfg, gp, dcfg, dp, z, noise = gan_setup(gan, batch)
    gen = _build(gcfg, gp, dcfg, dp, batch)
    img, mask = gen.generate_batch(z, noise)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    rgb_o, img_o, feats_o = o.generator(z, noise)
    logits_o, mask_o = o.decoder(feats_o)
    assert_same(img.cpu().numpy(), img_o, "%s image" % gan)
    assert_same(mask.cpu().numpy(), mask_o, "%s mask" % gan)
    assert 0.001 < mask_o.mean() < 0.999                     # the mask is not degenerate
    rgb, feats, img2 = gen.netG(z, noise=noise, want_image=True)
    assert_same(img2.cpu().numpy(), img_o, "%s image (two-call path)" % gan)
    assert_same(rgb.cpu().numpy(), rgb_o, "%s fp32 rgb" % gan)
    feats_np = [f.cpu().numpy() for f in feats]
    for i, (a, b) in enumerate(zip(feats_np, feats_o)):
        assert_same(a, b, "%s fp32 feature %d" % (gan, i))
    logits, mask2 = gen._decoder(*feats, want_mask=True)
    logits_np = logits.cpu().numpy()
    assert_same(logits_np, logits_o, "%s fp32 logits" % gan)
    assert_same(mask2.cpu().numpy(), mask_o, "%s mask (two-call path)" % gan)
    want = golden_bench_outputs().get("%s_b%d_fp32" % (gan, batch))
    if gan in ("ffhq", "cars"):
        assert want is not None, "tests/golden/bench_outputs.json lacks the fp32 digests of %s" % gan

        def h(a):
            return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        assert h(rgb.cpu().numpy()) == want["rgb_f32"] and h(logits_np) == want["logits_f32"]
        assert h(feats_np[-1]) == want["feature_last_f32"] and h(feats_np[-2]) == want["feature_second_last_f32"]
        assert h(img.cpu().numpy()) == want["image_u8"] and h(mask.cpu().numpy()) == want["mask_u8"]


def test_cli_generate_writes_dataset(torch_cuda, tmp_path):
    """`main.py generate` (reference main.py:75-104): same config.yml keys, img_%06d.jpg (RGB) and
    mask_%06d.png (single channel, class index) under BASE_DIR/dataset/train_generated."""
    import yaml
    from PIL import Image
    from gan_segmentation_amd import main as cli
    from gan_segmentation_amd import params as P
    from gan_segmentation_amd import weights as W
    gcfg = W.generator_config(8)                             # bedrooms, 256 px
    dcfg = W.decoder_config(8)
    gan_dir, base = tmp_path / "stylegan-models", tmp_path / "exp"
    gan_dir.mkdir()
    (base / "checkpoints").mkdir(parents=True)
    P.save_params(str(gan_dir / "stylegan-bedrooms.params"),
                  W.generator_names_to_scheme_s(W.synthetic_generator_params(gcfg)))   # structural names
    P.save_params(str(base / "checkpoints" / "checkpoint_last.params"), W.synthetic_decoder_params(dcfg))
    cfg = {"BASE_DIR": str(base), "GAN": "bedrooms", "GAN_DIR": str(gan_dir), "GAN_GPU_IDS": [0],
           "GAN_BATCH_SIZE_PER_GPU": 2, "SOLVER_GPU_IDS": [0], "ANNOTATION": "segmentation", "GENERATE_NUM": 3}
    (tmp_path / "config.yml").write_text(yaml.safe_dump(cfg))
    assert cli.main(["generate", "--config", str(tmp_path / "config.yml")]) == 0
    out = base / "dataset" / "train_generated"
    names = sorted(p.name for p in out.iterdir())
    assert names == ["img_000000.jpg", "img_000001.jpg", "img_000002.jpg",
                     "mask_000000.png", "mask_000001.png", "mask_000002.png"]
    img = Image.open(out / "img_000001.jpg")
    mask = np.asarray(Image.open(out / "mask_000001.png"))
    assert img.size == (256, 256) and img.mode == "RGB"
    assert mask.shape == (256, 256) and mask.dtype == np.uint8 and set(np.unique(mask)) <= {0, 1}
    # `annotation --count N`: the img/feat files the annotator GUI saves, loadable by the few-shot reader
    assert cli.main(["annotation", "--config", str(tmp_path / "config.yml")]) == 2            # no GUI here
    assert cli.main(["annotation", "--config", str(tmp_path / "config.yml"), "--count", "3"]) == 0
    from gan_segmentation_amd import annotation_io
    m_, img_a, feats = annotation_io.load_sample(str(base / "data"), 2)
    assert m_ is None and img_a.shape == (256, 256, 3) and len(feats) == 7 and feats[-1].shape == (64, 256, 256)
    # without a decoder checkpoint the reference prints "train Decoder first!" and exits -1
    (base / "checkpoints" / "checkpoint_last.params").unlink()
    assert cli.main(["generate", "--config", str(tmp_path / "config.yml")]) == -1


def test_config4_bedrooms_batch64_and_workspace_growth(torch_cuda):
    """BASELINE config 4 (bedrooms 256 px, batch 64) runs, and a sample's bytes do not depend on the
    batch it rides in (the batch-3 result is the one checked bit-exact against the oracle above);
    the workspace grows when a larger batch follows a smaller one."""
    from tests.common import gan_setup
    gcfg, gp, dcfg, dp, z, noise = gan_setup("bedrooms", 64)
    gen = _build(gcfg, gp, dcfg, dp, 64)
    i3, m3 = gen.generate_batch(z[:3], [a[:3] for a in noise])          # reserves 3
    i64, m64 = gen.generate_batch(z, noise)                              # grows to 64
    i1, m1 = gen.generate_batch(z[63:], [a[63:] for a in noise])
    assert_same(i64[:3].cpu().numpy(), i3.cpu().numpy(), "batch-64 vs batch-3 image")
    assert_same(m64[:3].cpu().numpy(), m3.cpu().numpy(), "batch-64 vs batch-3 mask")
    assert_same(i64[63:].cpu().numpy(), i1.cpu().numpy(), "last sample image")
    assert_same(m64[63:].cpu().numpy(), m1.cpu().numpy(), "last sample mask")
    assert 0.001 < m64.float().mean().item() < 0.999


_RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29733")
import numpy as np, torch, torch.distributed as dist
from gan_segmentation_amd import dist as gdist
from tests.common import reduced_setup
from gan_segmentation_amd.image_generator import ImageGenerator
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)           # RCCL, one rank: the collective call path of bench.py --gpus N
gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=3)
gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=3)
ref_img, ref_mask = gen.generate_batch(z, noise)
gat = gdist.PairGatherer(3, 128, 3, device=torch.device("cuda", 0), depth=2, force_collective=True)
for k in range(4):                                             # overlapped: wait only when a slot comes round again
    slot = k & 1
    gat.wait(slot)
    gen.generate_batch(z, noise, out=gat.buffers(slot))
    gat.submit(slot)
gat.wait_all()
torch.cuda.synchronize()
for slot in (0, 1):
    (img, mask), = gat.result(slot)
    assert torch.equal(img, ref_img) and torch.equal(mask, ref_mask)
gi, gm = gdist.gather_pairs(ref_img, ref_mask)                 # the blocking form degenerates to identity at one rank
assert gi is ref_img
dist.barrier(); dist.destroy_process_group()
print("RCCL_GATHER_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pair_gatherer_over_rccl_single_rank(torch_cuda, tmp_path):
    """The asynchronous gather of bench.py (dist.PairGatherer) through RCCL itself -- a one-rank group on this GPU:
    async gather into views of one preallocated tensor, stream-ordered wait, results identical to the direct call."""
    import subprocess
    import sys
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_GATHER_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


@pytest.mark.parametrize("fmap_max,dec_features,classes,batch", [(48, [48, 32, 48, 80, 16, 48], 2, 3), (96, [16, 96, 32, 80, 48, 16], 3, 2), (48, [32, 16, 32, 64, 128, 16], 2, 2)])
def test_odd_channel_counts_bit_exact(torch_cuda, oracle_lib, fmap_max, dec_features, classes, batch):
    """Channel counts that are multiples of 16 but not powers of two (3 or 6 output-channel groups, 5 channel
    blocks), a 3-class decoder and identity / 1x1 shortcuts in unusual places: every kernel selection path
    (tile geometry, channel tile, resident / streamed weights) must still reproduce the oracle bit for bit."""
    from gan_segmentation_amd import weights as W
    gcfg = W.generator_config(max_res_log2=7, fmap_base=3072, fmap_max=fmap_max)
    chans = W.generator_channels(gcfg)
    assert all(c % 16 == 0 for c in chans), chans
    dcfg = W.decoder_config(7, num_classes=classes, in_channels=chans)
    dcfg["features"] = list(dec_features) + [classes]
    gp = W.synthetic_generator_params(gcfg, seed=11, trivial_norm=False)
    dp = W.synthetic_decoder_params(dcfg, seed=12)
    z, noise = W.synthetic_inputs(gcfg, batch)
    gen = _build(gcfg, gp, dcfg, dp, batch)
    img, mask = gen.generate_batch(z, noise)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    img_o, mask_o = o.generate(z, noise)
    assert_same(img.cpu().numpy(), img_o, "image")
    assert_same(mask.cpu().numpy(), mask_o, "mask")
    assert mask_o.max() <= classes - 1


@pytest.mark.parametrize("start_res,use_bn", [(2, True), (5, True), (0, False), (1, False)])
def test_decoder_start_res_bit_exact(torch_cuda, oracle_lib, start_res, use_bn):
    """cfg['start_res'] (reference networks_seg.py:56,64,81,102): levels below it have no blocks, the first consumed
    feature is not concatenated; 5 = the final level alone.  The fused generate path and the decoder entry alone."""
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.networks_seg import Decoder
    gcfg = W.reduced_generator_config(7)
    gp = W.synthetic_generator_params(gcfg, seed=2, trivial_norm=False)
    dcfg = W.decoder_config(7, in_channels=W.generator_channels(gcfg))
    dcfg["start_res"], dcfg["use_bn"] = start_res, use_bn   # use_bn=False: no BatchNorm layers (networks_seg.py:20-33)
    dp = W.synthetic_decoder_params(dcfg, seed=3)
    z, noise = W.synthetic_inputs(gcfg, 2)
    gen = _build(gcfg, gp, dcfg, dp, 2)
    img, mask = gen.generate_batch(z, noise)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    img_o, mask_o = o.generate(z, noise)
    assert_same(img.cpu().numpy(), img_o, "image")
    assert_same(mask.cpu().numpy(), mask_o, "mask")
    _rgb, _img, feats = o.generator(z, noise)
    logits_o, mask_o2 = o.decoder(feats)
    dec = Decoder(dcfg, 1)
    dec.load_parameters(dp)
    logits, mask2 = dec(*feats, want_mask=True)
    assert_same(logits.cpu().numpy(), logits_o, "logits")
    assert_same(mask2.cpu().numpy(), mask_o2, "mask (decoder entry)")


@pytest.mark.parametrize("batch", [8, 4])
def test_benchmarked_config_ffhq(torch_cuda, batch):
    """What bench.py times -- BASELINE.json configs[1] (ffhq 1024^2, batch 8) and the per-GPU share of configs[2]
    (batch 4) on bench.py's own inputs.  Kernel selection depends on the batch (tile geometry, persistent forms,
    the stream-overlap rule flips at 8), so these batch sizes are checked themselves: the first samples against the
    C oracle's digests (tests/golden/bench_outputs.json), every sample against what batches of 1 and 2 produce
    (batch composition), with the decoder-beside-synthesis overlap off, on and by the default rule."""
    from tests.common import bench_setup, golden_bench_outputs, pair_digest
    gcfg, gp, dcfg, dp, z, noise = bench_setup("ffhq", batch)
    gen = _build(gcfg, gp, dcfg, dp, batch)
    ctx = gen.netG._model.ctx
    nlev = gcfg["max_res_log2"] - 1
    res = {}
    for levels in (-1, 0, nlev - 1):
        ctx.set_overlap(levels)
        img, mask = gen.generate_batch(z, noise)
        res[levels] = (img.cpu().numpy(), mask.cpu().numpy())
    ctx.set_overlap(-1)
    img, mask = res[-1]
    for levels in (0, nlev - 1):
        assert_same(res[levels][0], img, "image, overlap=%d" % levels)
        assert_same(res[levels][1], mask, "mask, overlap=%d" % levels)
    want = golden_bench_outputs()["ffhq_b%d" % batch]["samples"]
    for i, h in enumerate(want):
        assert pair_digest(img[i], mask[i]) == h, "sample %d of the batch-%d run differs from the oracle" % (i, batch)
    lo = 0
    for size in [1, 2] * batch:
        hi = min(batch, lo + size)
        i2, m2 = gen.generate_batch(z[lo:hi], [a[lo:hi] for a in noise])
        assert_same(i2.cpu().numpy(), img[lo:hi], "image of samples %d:%d alone" % (lo, hi))
        assert_same(m2.cpu().numpy(), mask[lo:hi], "mask of samples %d:%d alone" % (lo, hi))
        lo = hi
        if lo == batch:
            break
    assert 0.001 < mask.mean() < 0.999


def test_bench_secondary_configs_match_oracle_digests(torch_cuda):
    """bench.py's secondary fp32 measurement (BASELINE.json configs[3]: bedrooms 256^2, batch 64) on its own inputs:
    the first samples equal the C oracle's digests."""
    from tests.common import bench_setup, golden_bench_outputs, pair_digest
    gcfg, gp, dcfg, dp, z, noise = bench_setup("bedrooms", 64)
    gen = _build(gcfg, gp, dcfg, dp, 64)
    img, mask = gen.generate_batch(z, noise)
    img, mask = img.cpu().numpy(), mask.cpu().numpy()
    for i, h in enumerate(golden_bench_outputs()["bedrooms_b64"]["samples"]):
        assert pair_digest(img[i], mask[i]) == h, "bedrooms sample %d" % i


@pytest.mark.parametrize("gan", ["bedrooms", "cars", "ffhq"])
def test_full_size_semantic_tolerance(torch_cuda, gan):
    """The north-star sentence as a test, at full size: against the INDEPENDENT reference-order restatement
    (oracle/ref_semantic.py: torch functionals, the reference's 9-tap convolutions, NCHW, two-pass instance norm)
    max |rgb diff| <= 1e-3 and max |logit diff| <= 1e-3 fp32, masks equal wherever the two logits are further
    apart than 1e-3 (reference image_generator.py:86-124, seg_solver.py:307-329)."""
    from oracle import ref_semantic as S
    from tests.common import gan_setup
    gcfg, gp, dcfg, dp, z, noise = gan_setup(gan, 1)
    gen = _build(gcfg, gp, dcfg, dp, 1)
    rgb, feats, img = gen.netG(z, noise=noise, want_image=True)
    logits, mask = gen._decoder(*feats, want_mask=True)
    simg, smask, srgb, sfeats, slog = S.generate(gcfg, gp, dcfg, dp, z, noise)
    rgb, logits, mask = rgb.cpu().numpy(), logits.cpu().numpy(), mask.cpu().numpy()
    assert np.abs(rgb - srgb).max() <= 1e-3, np.abs(rgb - srgb).max()
    assert np.abs(logits - slog).max() <= 1e-3, np.abs(logits - slog).max()
    margin = np.abs(slog[:, 1] - slog[:, 0])
    differ = mask != smask
    assert not (differ & (margin > 1e-3)).any()
    assert differ.mean() < 1e-4                                 # near-ties are rare
    assert np.abs(img.cpu().numpy().astype(int) - simg.astype(int)).max() <= 1   # u8 truncation of a 1e-3-close value
    for f, sf in zip(feats, sfeats):
        assert np.abs(f.cpu().numpy() - sf).max() <= 1e-3 * max(1.0, np.abs(sf).max())



def test_semantic_tolerance_of_the_benchmarked_batch(torch_cuda):
    """The independent check at the BENCHMARKED batch size (kernel selection varies with the batch: persistent / one-tile
    forms, channel tiles): sample 7 of bench.py's ffhq batch of 8 against the reference-order restatement evaluated on that
    sample alone (reference image_generator.py:86-124, seg_solver.py:307-329)."""
    from oracle import ref_semantic as S
    from tests.common import bench_setup
    gcfg, gp, dcfg, dp, z, noise = bench_setup("ffhq", 8)
    gen = _build(gcfg, gp, dcfg, dp, 8)
    rgb, feats, _img = gen.netG(z, noise=noise, want_image=True)
    logits, mask = gen._decoder(*feats, want_mask=True)
    k = 7
    _simg, smask, srgb, _sf, slog = S.generate(gcfg, gp, dcfg, dp, z[k:k + 1], [a[k:k + 1] for a in noise])
    rgb, logits, mask = rgb[k:k + 1].cpu().numpy(), logits[k:k + 1].cpu().numpy(), mask[k:k + 1].cpu().numpy()
    assert np.abs(rgb - srgb).max() <= 1e-3, np.abs(rgb - srgb).max()
    assert np.abs(logits - slog).max() <= 1e-3, np.abs(logits - slog).max()
    margin = np.abs(slog[:, 1] - slog[:, 0])
    assert not ((mask != smask) & (margin > 1e-3)).any()


@pytest.mark.parametrize("gan", ["bedrooms", "cars", "ffhq"])
def test_accuracy_against_the_fp64_yardstick(torch_cuda, gan):
    """A yardstick for the 1e-3 tolerance: tests/golden/f64_yardstick.npz holds the reference-order restatement evaluated in
    FLOAT64 (strided samples, tests/golden/make_f64_yardstick.py) and the error of the same restatement in fp32.  The HIP path
    -- Winograd F(2x2,3x3) and F(2x2,2x2) forms, sub-pixel form, K split, AdaIN folded into one fma -- must be no further from the
    float64 result than THREE TIMES the fp32 reference order is (measured: 2.4-2.7x on rgb, 1.9-2.7x on logits; absolute 2.6e-5 /
    3.9e-5 on an rgb range of +-4.2 / +-5.2).  The factor is NOT the Winograd or sub-pixel forms -- with both switched off the C
    oracle is 2.3x / 3.1x away (tests/test_oracle.py::test_winograd_forms_are_as_accurate_as_the_direct_chains) -- it is the
    matrix core's single k-ordered fmaf chain over up to 4608 terms against the CPU library's blocked partial sums."""
    from tests.common import gan_setup
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f64_yardstick.npz"))
    st = int(g[gan + "_stride"])
    gcfg, gp, dcfg, dp, z, noise = gan_setup(gan, 1)
    gen = _build(gcfg, gp, dcfg, dp, 1)
    rgb, feats, _img = gen.netG(z, noise=noise, want_image=True)
    logits, _mask = gen._decoder(*feats, want_mask=True)
    e_rgb = np.abs(rgb.cpu().numpy().astype(np.float64)[:, :, ::st, ::st] - g[gan + "_rgb"]).max()
    e_log = np.abs(logits.cpu().numpy().astype(np.float64)[:, :, ::st, ::st] - g[gan + "_logits"]).max()
    ref_rgb, ref_log = float(g[gan + "_sem32_err_rgb"]), float(g[gan + "_sem32_err_logits"])
    assert e_rgb <= 3.0 * ref_rgb, "rgb: HIP %.3g vs fp32 reference order %.3g from the float64 result" % (e_rgb, ref_rgb)
    assert e_log <= 3.0 * ref_log, "logits: HIP %.3g vs fp32 reference order %.3g from the float64 result" % (e_log, ref_log)
    assert e_rgb <= 1e-4 and e_log <= 1e-4           # a tenth of the north-star tolerance


def test_repeated_calls_are_byte_identical(torch_cuda):
    """The step exchanges data between workgroups inside kernels (tagged words of the mapping network, 64-bit atomic
    statistic rows that the finalize kernel clears again): the same inputs must give the same bytes call after call,
    with another model's calls in between, and after a change of batch size."""
    import torch
    from tests.common import pair_digest
    a = reduced_setup(7, batch=3, seed=4)
    b = reduced_setup(6, batch=2, seed=6)
    ga = _build(a[0], a[1], a[2], a[3], 3)
    gb = _build(b[0], b[1], b[2], b[3], 2)
    first = {}
    for it in range(12):
        for key, gen, st in (("a", ga, a), ("b", gb, b)):
            n = len(st[4]) if it % 3 else 1                     # every third round a one-sample call in between
            img, mask = gen.generate_batch(st[4][:n], [x[:n] for x in st[5]])
            torch.cuda.synchronize()
            d = pair_digest(img.cpu().numpy(), mask.cpu().numpy())
            assert first.setdefault((key, n), d) == d, "call %d of model %s (n=%d) differs from the first one" % (it, key, n)
    assert len(first) == 4


def test_two_live_models_do_not_share_state(torch_cuda, oracle_lib):
    """Every Generator / Decoder owns its context (the reference's gluon blocks are independent objects): two models
    of DIFFERENT configurations and two of the same configuration with different weights stay bit-exact against
    their own oracle while their calls interleave on one GPU."""
    from gan_segmentation_amd import weights as W
    a = reduced_setup(7, batch=2, seed=2)
    b = reduced_setup(6, batch=3, seed=5)
    c = reduced_setup(7, batch=2, seed=9)          # same configuration as `a`, other weights
    gens = [_build(s[0], s[1], s[2], s[3], len(s[4])) for s in (a, b, c)]
    want = [oracle_lib.Oracle(s[0], s[1], s[2], s[3]).generate(s[4], s[5]) for s in (a, b, c)]
    assert not np.array_equal(want[0][0], want[2][0])
    for order in ((0, 1, 2), (2, 1, 0), (1, 0, 2)):
        outs = {}
        for k in order:
            outs[k] = gens[k].generate_batch(setups_z(k, a, b, c), setups_noise(k, a, b, c))
        for k in order:
            assert_same(outs[k][0].cpu().numpy(), want[k][0], "image of model %d" % k)
            assert_same(outs[k][1].cpu().numpy(), want[k][1], "mask of model %d" % k)
    # the reference surface too: an independent Generator per configuration, called alternately
    from gan_segmentation_amd.networks_stylegan import Generator
    ga, gb = Generator(a[0]), Generator(b[0])
    ga.load_parameters(a[1])
    gb.load_parameters(b[1])
    oa, ob = oracle_lib.Oracle(a[0], a[1]), oracle_lib.Oracle(b[0], b[1])
    for _ in range(2):
        rgb_a, _fa = ga(a[4], noise=a[5])
        rgb_b, _fb = gb(b[4], noise=b[5])
        assert_same(rgb_a.cpu().numpy(), oa.generator(a[4], a[5])[0], "rgb of generator a")
        assert_same(rgb_b.cpu().numpy(), ob.generator(b[4], b[5])[0], "rgb of generator b")
    assert W.generator_channels(a[0]) != W.generator_channels(b[0])


def setups_z(k, *setups):
    return setups[k][4]


def setups_noise(k, *setups):
    return setups[k][5]


def test_pointer_array_lengths_are_validated(torch_cuda):
    """The C ABI takes the entry counts of its pointer arrays: arrays sized for another configuration are refused
    (GSA_ERR_INVALID) instead of being read out of bounds."""
    import torch
    from gan_segmentation_amd import _lib
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=1)
    gen = _build(gcfg, gp, dcfg, dp, 1)
    gen.generate_batch(z, noise)                                 # reserves the workspace
    ctx = gen.netG._model.ctx
    dev = gen.netG._model.device
    zt = torch.from_numpy(z).to(dev)
    nz = [torch.from_numpy(a).to(dev) for a in noise]
    img = torch.empty((1, 128, 128, 3), dtype=torch.uint8, device=dev)
    mask = torch.empty((1, 128, 128), dtype=torch.uint8, device=dev)
    with pytest.raises(_lib.GsaError, match="noise planes"):
        ctx.generate(None, 1, zt.data_ptr(), [a.data_ptr() for a in nz[:-2]], img.data_ptr(), mask.data_ptr())
    with pytest.raises(_lib.GsaError, match="feature pointers"):
        ctx.decoder_forward(None, 1, [nz[0].data_ptr()] * 3, None, mask.data_ptr())
    with pytest.raises(_lib.GsaError, match="feature pointers"):
        ctx.generator_forward(None, 1, zt.data_ptr(), [a.data_ptr() for a in nz], None, img.data_ptr(), [nz[0].data_ptr()] * 2)


def test_in_process_device_list(torch_cuda, oracle_lib, tmp_path):
    """``ImageGenerator(gpu_ids=[a, b])`` / ``SegSolver(gpu_ids=[a, b])``: the reference's in-process device list
    (image_generator.py:17,95-101; seg_solver.py:24-28,317-325) -- one replica (context) per entry, the batch split
    like split_and_load(even_split=False), results in sample order.  One GPU here, so both entries name device 0:
    two contexts on two streams' worth of launches, same bytes as the single replica and the oracle."""
    from gan_segmentation_amd import params as P
    from gan_segmentation_amd.image_generator import ImageGenerator
    from gan_segmentation_amd.seg_solver import SegSolver
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=5)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    img_o, mask_o = o.generate(z, noise)
    gen2 = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0, 0], batch_size=5)
    assert len(gen2._gens) == 2 and gen2._gens[0]._model is not gen2._gens[1]._model
    img, mask = gen2.generate_batch(z, noise)                    # 3 + 2 samples
    assert_same(img.cpu().numpy(), img_o, "image over two replicas")
    assert_same(mask.cpu().numpy(), mask_o, "mask over two replicas")
    i1, m1 = gen2.generate_batch(z[:1], [a[:1] for a in noise])  # fewer samples than replicas
    assert_same(i1.cpu().numpy(), img_o[:1], "one sample over two replicas")
    _rgb_o, imgs_o, feats_o = o.generator(z, noise)
    got = list(gen2.get_images(5, latents=z, noise=noise))
    assert len(got) == 5
    ckpt = tmp_path / "checkpoints"
    ckpt.mkdir()
    P.save_params(str(ckpt / "checkpoint_last.params"), dp)
    solver = SegSolver(7, str(tmp_path / "data"), str(ckpt), gpu_ids=[0, 0], keep_weights=False, in_channels=dcfg["in_channels"])
    assert solver.is_trained and len(solver.nets) == 2
    for i, (im, feats) in enumerate(got):
        assert_same(im, imgs_o[i], "get_images image %d" % i)
        for f, fo in zip(feats, feats_o):
            assert_same(f, fo[i], "get_images feature")
    m = solver.predict([fo for fo in feats_o])                   # 4-D features, batch 5 split 3 + 2
    assert m.shape == (5, 128, 128, 1)
    assert_same(m[..., 0].astype(np.uint8), mask_o, "predict over two replicas")
    # indexed generation is sharding-independent: two replicas == one
    gen1 = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=5)
    ia, ma = gen1.generate_indexed(40, 5, seed=3)
    ib, mb = gen2.generate_indexed(40, 5, seed=3)
    assert_same(ia.cpu().numpy(), ib.cpu().numpy(), "indexed image")
    assert_same(ma.cpu().numpy(), mb.cpu().numpy(), "indexed mask")


def test_large_activations_stay_bit_exact(torch_cuda, oracle_lib):
    """The instance-norm statistics are 64-bit fixed-point sums (include/gsa.h states the range: |x| < 2.3e4 per value,
    rms < 2.9e3 over a 1024^2 plane).  Activations three orders of magnitude above the synthetic weights' O(1) -- one level's
    conv weights scaled so that its pre-normalisation tensor has rms ~ 1e3 -- are still inside it: same bits as the oracle."""
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=2, trivial_norm=False)
    gp = dict(gp)
    for name in ("64_conv_2_weight", "128_deconv_1_weight", "16_conv_1_weight"):
        gp[name] = (np.asarray(gp[name], np.float32) * np.float32(800.0)).astype(np.float32)
    gen = _build(gcfg, gp, dcfg, dp, 2)
    img, mask = gen.generate_batch(z, noise)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    img_o, mask_o = o.generate(z, noise)
    assert_same(img.cpu().numpy(), img_o, "image")
    assert_same(mask.cpu().numpy(), mask_o, "mask")



@pytest.mark.parametrize("scale", [1e-3, 1e-4])
@pytest.mark.parametrize("where", ["reduced", "bedrooms"])
def test_small_activations_stay_within_the_tolerance(torch_cuda, where, scale):
    """The LOW end of the statistics' range, judged against the INDEPENDENT reference-order restatement
    (oracle/ref_semantic.py: two-pass instance norm) -- HIP == C oracle holds regardless because both share the rule.  One
    level's conv weights scaled by 1e-3 / 1e-4: its pre-normalisation tensor is then noise*scale + bias with a spread far
    below the mean, the case where E[x^2] - mean^2 loses what a coarse fixed-point unit rounds away (at the round-3 unit 2^-20
    a 4x4 plane came out 1e-2 off in rgb).  With the unit 2^-S2, S2 = clamp(40 - log2(HW), 20, 26) (include/gsa.h) the
    north-star 1e-3 holds at every level (reference networks_stylegan.py:239-264)."""
    from oracle import ref_semantic as S
    from tests.common import gan_setup
    if where == "reduced":
        gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=2, trivial_norm=False)
        names = ["4_conv_2_weight", "8_conv_1_weight", "16_conv_2_weight", "32_conv_2_weight", "64_conv_1_weight", "128_deconv_1_weight"]
    else:
        gcfg, gp, dcfg, dp, z, noise = gan_setup("bedrooms", 1)
        names = ["4_conv_2_weight", "8_conv_2_weight", "64_conv_1_weight", "256_conv_2_weight"]
    n = len(z)
    for name in names:
        g2 = dict(gp)
        g2[name] = (np.asarray(gp[name], np.float32) * np.float32(scale)).astype(np.float32)
        gen = _build(gcfg, g2, dcfg, dp, n)
        rgb, feats, _img = gen.netG(z, noise=noise, want_image=True)
        logits, mask = gen._decoder(*feats, want_mask=True)
        _simg, smask, srgb, sfeats, slog = S.generate(gcfg, g2, dcfg, dp, z, noise)
        e_rgb, e_log = np.abs(rgb.cpu().numpy() - srgb).max(), np.abs(logits.cpu().numpy() - slog).max()
        assert e_rgb <= 1e-3 and e_log <= 1e-3, "%s x%g: rgb %.3g logits %.3g" % (name, scale, e_rgb, e_log)
        margin = np.abs(slog[:, 1] - slog[:, 0])
        assert not ((mask.cpu().numpy() != smask) & (margin > 1e-3)).any(), name
        for f, sf in zip(feats, sfeats):
            assert np.abs(f.cpu().numpy() - sf).max() <= 1e-3 * max(1.0, np.abs(sf).max()), name
        gen.netG._model.ctx.check()
        del gen


def test_large_activations_stay_within_the_tolerance(torch_cuda):
    """The UPPER end against the independent restatement too: three levels scaled by 800 (pre-normalisation rms ~1e3, single
    values beyond the 2^(50-S2) quad limit of the fine unit: those quads take the two-level conversion) -- still <= 1e-3."""
    from oracle import ref_semantic as S
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=2, trivial_norm=False)
    gp = dict(gp)
    for name in ("64_conv_2_weight", "128_deconv_1_weight", "16_conv_1_weight"):
        gp[name] = (np.asarray(gp[name], np.float32) * np.float32(800.0)).astype(np.float32)
    gen = _build(gcfg, gp, dcfg, dp, 2)
    rgb, feats, _img = gen.netG(z, noise=noise, want_image=True)
    logits, mask = gen._decoder(*feats, want_mask=True)
    _simg, smask, srgb, _sf, slog = S.generate(gcfg, gp, dcfg, dp, z, noise)
    assert np.abs(rgb.cpu().numpy() - srgb).max() <= 1e-3 and np.abs(logits.cpu().numpy() - slog).max() <= 1e-3
    gen.netG._model.ctx.check()


def test_device_side_checks_are_reported(torch_cuda, oracle_lib, monkeypatch):
    """The silent-failure holes of the stream-ordered path (include/gsa.h gsa_check), each with a forced condition:
    (a) the fused mapping network's exchange times out (one workgroup short: gsa_debug_inject kind 1) -> GSA_ERR_DEVICE;
    (b) an instance-norm statistic leaves its fixed-point range (one level's weights scaled by 1e4) -> GSA_ERR_DEVICE;
    (c) a pass that fails half way (an injected error between a statistics producer and its finalize: kind 2; a null
        noise plane at level 3) leaves no dirty statistic rows behind: the next pass on the same context is still bit-exact."""
    from gan_segmentation_amd._lib import GsaError
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=2, trivial_norm=False)
    img_o, mask_o = oracle_lib.Oracle(gcfg, gp, dcfg, dp).generate(z, noise)
    # (a)
    monkeypatch.setenv("GSA_FAULT", "1")            # the environment arms nothing any more (round 4): only the explicit call does
    bad = _build(gcfg, gp, dcfg, dp, 2)
    monkeypatch.delenv("GSA_FAULT")
    monkeypatch.delenv("GSA_TEST_HOOKS")            # round 5: and the explicit call alone neither -- it needs GSA_TEST_HOOKS=1 as well
    with pytest.raises(GsaError, match="test hooks are off"):
        bad.netG._model.ctx.debug_inject(1)
    monkeypatch.setenv("GSA_TEST_HOOKS", "1")
    bad.netG._model.ctx.debug_inject(1)
    with pytest.raises(GsaError, match="mapping network timed out"):
        bad.generate_batch(z, noise)                 # the shim checks after a context's first step
    bad.netG._model.ctx.check()                      # reported once, then clear again
    del bad
    # (b)
    gp_big = dict(gp)
    gp_big["128_conv_2_weight"] = (np.asarray(gp["128_conv_2_weight"], np.float32) * np.float32(1e4)).astype(np.float32)
    big = _build(gcfg, gp_big, dcfg, dp, 2)
    with pytest.raises(GsaError, match="instance-norm statistic"):
        big.generate_batch(z, noise)
    del big
    # (c) first an injected failure between a producer and its finalize (dirty rows), then an argument error half way
    gen = _build(gcfg, gp, dcfg, dp, 2)
    gen.netG._model.ctx.debug_inject(2)
    with pytest.raises(GsaError, match="injected fault"):
        gen.generate_batch(z, noise)
    img, mask = gen.generate_batch(z, noise)
    assert_same(img.cpu().numpy(), img_o, "image after the injected fault")
    assert_same(mask.cpu().numpy(), mask_o, "mask after the injected fault")
    holes = [torch_cuda.from_numpy(a).cuda() for a in noise]
    zd = torch_cuda.from_numpy(z).cuda()
    ctx = gen.netG._model.ctx
    img_t = torch_cuda.empty((2, 128, 128, 3), dtype=torch_cuda.uint8, device="cuda")
    mask_t = torch_cuda.empty((2, 128, 128), dtype=torch_cuda.uint8, device="cuda")
    ptrs = [a.data_ptr() for a in holes]
    ptrs[6] = None                                    # level 3, first plane: the levels before it have launched
    with pytest.raises(GsaError, match="noise plane 6 is null"):
        ctx.generate(torch_cuda.cuda.current_stream().cuda_stream, 2, zd.data_ptr(), ptrs, img_t.data_ptr(), mask_t.data_ptr())
    img2, mask2 = gen.generate_batch(z, noise)
    assert_same(img2.cpu().numpy(), img_o, "image after the failing call")
    assert_same(mask2.cpu().numpy(), mask_o, "mask after the failing call")
    ctx.check()
    assert "hipcc" in ctx.api.version().decode() and "clang" in ctx.api.version().decode()



def test_graph_replay_of_small_steps_matches_the_oracle(torch_cuda, oracle_lib, monkeypatch):
    """`generate_batch` replays a captured hipGraph when the very same call repeats (small batches / bf16 mode; GSA_GRAPH=1 forces
    it): the replay must read the CURRENT contents of the input tensors and write the current outputs -- same bytes as the
    eager call and as the oracle, also after the inputs were rewritten in place and after the workspace was re-reserved."""
    monkeypatch.setenv("GSA_GRAPH", "1")
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=4, trivial_norm=False)
    o = oracle_lib.Oracle(gcfg, gp, dcfg, dp)
    want = [o.generate(z[k:k + 2], [a[k:k + 2] for a in noise]) for k in (0, 2)]
    gen = _build(gcfg, gp, dcfg, dp, 2)
    zt = torch_cuda.from_numpy(z[:2].copy()).cuda()
    nt = [torch_cuda.from_numpy(a[:2].copy()).cuda() for a in noise]
    out = (torch_cuda.empty((2, 128, 128, 3), dtype=torch_cuda.uint8, device="cuda"),
           torch_cuda.empty((2, 128, 128), dtype=torch_cuda.uint8, device="cuda"))
    model = gen.netG._model
    for it in range(40):                                 # the 32nd identical call is captured, the later ones are replays
        k = 0 if it % 2 == 0 else 2                      # the SAME tensors, rewritten in place: the key does not change
        zt.copy_(torch_cuda.from_numpy(z[k:k + 2].copy()))
        for t, a in zip(nt, noise):
            t.copy_(torch_cuda.from_numpy(a[k:k + 2].copy()))
        out[0].zero_(); out[1].zero_()
        img, mask = gen.generate_batch(zt, nt, out=out)
        assert_same(img.cpu().numpy(), want[k // 2][0], "image, call %d" % it)
        assert_same(mask.cpu().numpy(), want[k // 2][1], "mask, call %d" % it)
    assert len(model.__dict__.get("_graphs", {})) == 1, "the repeated call was never captured"
    # a larger batch re-reserves the workspace: the old graph's pointers are stale and must not be replayed
    gen4 = gen.generate_batch(z, noise)
    img, mask = gen.generate_batch(zt, nt, out=out)
    assert_same(img.cpu().numpy(), want[1][0], "image after re-reserve")
    assert gen4[0].shape[0] == 4
    monkeypatch.setenv("GSA_GRAPH", "0")
    img, mask = gen.generate_batch(zt, nt, out=out)
    assert_same(mask.cpu().numpy(), want[1][1], "mask, eager again")


def test_graph_replay_after_a_failed_pass_is_still_bit_exact(torch_cuda, oracle_lib):
    """a pass that dies between a statistics producer and its finalize leaves dirty rows which only the next
    EAGER pass re-zeroes; a cached hipGraph of the same call has no memsets.  A failing call therefore ends the graph epoch:
    the repeated key misses, runs eagerly (re-zeroing), is captured again -- same bytes as the oracle throughout."""
    from gan_segmentation_amd._lib import GsaError
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=2, trivial_norm=False)
    img_o, mask_o = oracle_lib.Oracle(gcfg, gp, dcfg, dp).generate(z, noise)
    gen = _build(gcfg, gp, dcfg, dp, 2)
    gen.graph_mode, gen.graph_after = "1", 3
    zt = torch_cuda.from_numpy(z).cuda()
    nt = [torch_cuda.from_numpy(a).cuda() for a in noise]
    out = (torch_cuda.empty((2, 128, 128, 3), dtype=torch_cuda.uint8, device="cuda"),
           torch_cuda.empty((2, 128, 128), dtype=torch_cuda.uint8, device="cuda"))
    model = gen.netG._model
    for _ in range(5):
        gen.generate_batch(zt, nt, out=out)
    assert gen.graphs_captured() == 1
    epoch = model.ctx.graph_epoch
    # a failing eager call of ANOTHER key (a null noise plane at level 3: the levels before it have launched their producers)
    ptrs = [a.data_ptr() for a in nt]
    ptrs[6] = None
    with pytest.raises(GsaError, match="noise plane 6 is null"):
        model.ctx.generate(torch_cuda.cuda.current_stream().cuda_stream, 2, zt.data_ptr(), ptrs, out[0].data_ptr(), out[1].data_ptr())
    model.ctx.debug_inject(2)                         # and one that dies between a producer and its finalize
    with pytest.raises(GsaError, match="injected fault"):
        model.ctx.generate(torch_cuda.cuda.current_stream().cuda_stream, 2, zt.data_ptr(), [a.data_ptr() for a in nt], out[0].data_ptr(), out[1].data_ptr())
    assert model.ctx.graph_epoch > epoch
    for it in range(5):                               # the old graph is never replayed; eager, then captured again
        out[0].zero_(); out[1].zero_()
        img, mask = gen.generate_batch(zt, nt, out=out)
        assert_same(img.cpu().numpy(), img_o, "image, call %d after the failed passes" % it)
        assert_same(mask.cpu().numpy(), mask_o, "mask, call %d after the failed passes" % it)
    assert gen.graphs_captured() == 2
    model.ctx.check()


def test_status_snapshot_with_an_idle_replica(torch_cuda):
    """ImageGenerator(gpu_ids=[a, b]) with ONE sample -- split_sizes drops the empty slice, so the second
    replica never reserved a workspace -- must still answer snapshot_status() (main.py generate calls it for every batch): the
    idle replica reports {0, 0}.  (Two contexts on device 0, as in test_in_process_device_list.)"""
    gcfg, gp, dcfg, dp, z, noise = reduced_setup(7, batch=1)
    from gan_segmentation_amd.image_generator import ImageGenerator
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0, 0], batch_size=2)
    img, mask = gen.generate_batch(z, noise)
    views = gen.snapshot_status()
    torch_cuda.cuda.synchronize()
    assert len(views) == 2 and all(int(v[0]) == 0 and int(v[1]) == 0 for v in views)
    assert img.shape[0] == 1 and mask.shape[0] == 1


def test_cli_generate_stops_at_the_first_bad_batch(torch_cuda, tmp_path, monkeypatch, capsys):
    """VERDICT r3 item 7 (reference main.py:93-104): the sticky device words travel with every batch (gsa_status_snapshot: 8
    bytes behind the batch's kernels, no sync) and the writer looks at batch k's copy before it releases batch k's files.  A
    6-batch CLI run whose statistics-range word is forced from batch 3 on: non-zero exit naming global index 6, the files
    of indices 0-5 on disk, none from 6 on."""
    import yaml
    from gan_segmentation_amd import main as cli
    from gan_segmentation_amd import params as P
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.image_generator import ImageGenerator
    gcfg = W.generator_config(8)
    dcfg = W.decoder_config(8)
    gan_dir, base = tmp_path / "stylegan-models", tmp_path / "exp"
    gan_dir.mkdir()
    (base / "checkpoints").mkdir(parents=True)
    P.save_params(str(gan_dir / "stylegan-bedrooms.params"), W.generator_names_to_scheme_s(W.synthetic_generator_params(gcfg)))
    P.save_params(str(base / "checkpoints" / "checkpoint_last.params"), W.synthetic_decoder_params(dcfg))
    cfg = {"BASE_DIR": str(base), "GAN": "bedrooms", "GAN_DIR": str(gan_dir), "GAN_GPU_IDS": [0],
           "GAN_BATCH_SIZE_PER_GPU": 2, "SOLVER_GPU_IDS": [0], "ANNOTATION": "segmentation", "GENERATE_NUM": 12}
    (tmp_path / "config.yml").write_text(yaml.safe_dump(cfg))
    real = ImageGenerator.generate_indexed
    armed = []

    def generate_indexed(self, first_index, n, seed=0, out=None):
        if not armed:                                  # before the first batch: 3 clean passes, the 4th (index 6) sets the word
            armed.append(1)
            self.netG._model.ctx.debug_inject(3, 3)
        return real(self, first_index, n, seed=seed, out=out)

    monkeypatch.setattr(ImageGenerator, "generate_indexed", generate_indexed)
    rc = cli.main(["generate", "--config", str(tmp_path / "config.yml")])
    err = capsys.readouterr().err
    assert rc == -2 and "global sample index 6" in err and "instance-norm statistic" in err, err
    names = sorted(p.name for p in (base / "dataset" / "train_generated").iterdir())
    assert names == sorted(["img_%06d.jpg" % i for i in range(6)] + ["mask_%06d.png" % i for i in range(6)])
    # a clean run of the same configuration exits 0 with all 24 files
    monkeypatch.setattr(ImageGenerator, "generate_indexed", real)
    assert cli.main(["generate", "--config", str(tmp_path / "config.yml")]) == 0
    assert len(list((base / "dataset" / "train_generated").iterdir())) == 24


_W43_WORKER = r'''
import sys
import numpy as np
sys.path.insert(0, ROOT_DIR)
from tests.common import reduced_setup, gan_setup
from gan_segmentation_amd.image_generator import ImageGenerator
from oracle.binding import Oracle
for name, setup in (("reduced", lambda: reduced_setup(7, batch=2, trivial_norm=False)), ("bedrooms", lambda: gan_setup("bedrooms", 2))):
    gcfg, gp, dcfg, dp, z, noise = setup()
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=2)
    rgb, feats, img = gen.netG(z, noise=noise, want_image=True)
    logits, mask = gen._decoder(*feats, want_mask=True)
    o = Oracle(gcfg, gp, dcfg, dp)
    orgb, oimg, ofeats = o.generator(z, noise)
    ologits, omask = o.decoder(ofeats)
    for i, (f, of) in enumerate(zip(feats, ofeats)):
        assert np.array_equal(f.cpu().numpy(), of), "%s feature %d" % (name, i)
    assert np.array_equal(rgb.cpu().numpy(), orgb) and np.array_equal(img.cpu().numpy(), oimg), name
    assert np.array_equal(logits.cpu().numpy(), ologits) and np.array_equal(mask.cpu().numpy(), omask), name
    # the stand-alone decoder (no AdaIN on its inputs: the kernel's no-affine path) on the exported features
    lg2, mk2 = gen._decoder(*[f.clone() for f in feats], want_mask=True)
    assert np.array_equal(lg2.cpu().numpy(), ologits), name
    gen.netG._model.ctx.check()
print("W43_OK")
'''


def test_winograd_f4x4_form_is_bit_exact_when_selected(torch_cuda, tmp_path):
    """conv3x3_wino43 -- Winograd F(4x4,3x3) on the matrix cores for the streamed-weight 3x3 layers (36 products
    per 16 outputs; 8-channel items, both operands by LDS-DMA) -- is OPT-IN (GSA_WINO43=1 with the oracle's GSAO_WINO43=1: it
    measured 15-20 % slower than F(2x2,3x3)).  Selected, it reproduces the oracle's F(4x4,3x3) arithmetic bit for
    bit: reduced 128 px config (g.32.conv_2: every tile an edge tile) and bedrooms 256^2 (conv_2 at 32^2-256^2, decoder cvt at
    64^2-256^2, interior tiles, both epilogues, the no-AdaIN path) -- reference networks_stylegan.py:354-457."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from gan_segmentation_amd import _lib
    if not os.path.exists(_lib.EXPERIMENTS_LIBRARY):
        pytest.skip("libgsa_hip_exp.so is not built (make -C gan-segmentation_amd/csrc experiments)")
    script = tmp_path / "w43_worker.py"
    script.write_text(_W43_WORKER.replace("ROOT_DIR", repr(root)))
    out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, GSA_WINO43="1", GSAO_WINO43="1", GSA_HIP_LIBRARY="libgsa_hip_exp.so"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "W43_OK" in out.stdout, out.stdout[-800:] + out.stderr[-2500:]


_SWITCH_WORKER = r'''
import sys
sys.path.insert(0, ROOT_DIR)
from tests.common import bench_setup, golden_bench_outputs, pair_digest
from gan_segmentation_amd.image_generator import ImageGenerator
gcfg, gp, dcfg, dp, z, noise = bench_setup("ffhq", 4)
gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=4)
img, mask = gen.generate_batch(z, noise)
img, mask = img.cpu().numpy(), mask.cpu().numpy()
assert pair_digest(img[0], mask[0]) == golden_bench_outputs()["ffhq_b4"]["samples"][0]
print("SWITCH_OK")
'''


_OLD = {"GSA_WINO_LEAN": "0", "GSA_SUB_LEAN": "0", "GSA_POST_PK": "0", "GSA_POST_DMA": "0"}      # the kernels of rounds 1-4 (the fallbacks): their own switches only act there
_EXP = {"GSA_HIP_LIBRARY": "libgsa_hip_exp.so"}                               # `make experiments`: + the measured-slower kernels of round 4


@pytest.mark.parametrize("env", [
    # round 5: the lean kernels off / partly on, the single-buffered three-workgroups-per-CU form, the packed post pass and final conv
    _OLD, {"GSA_WINO_LEAN": "1"}, {"GSA_WINO_LEAN": "3"}, {"GSA_SUB_LEAN": "0"}, {"GSA_POST_PK": "0"}, {"GSA_POST_PK": "2"}, {"GSA_FINAL_PK": "0"},
    {"GSA_WINO_LEAN_SB": "1"}, {"GSA_WINO_LEAN_PF": "2"}, {"GSA_FUSE_RGB": "0"}, {"GSA_POST_DMA": "0"}, {"GSA_POST_DMA_OST": "0"}, {"GSA_POST_DMA_BH": "16"},
    # rounds 2-4, on the fallback kernels they belong to
    dict(_OLD, GSA_WINO_NT="2"), dict(_OLD, GSA_WINO_GM="0"), dict(_OLD, GSA_WINO_CHUNK="1"), dict(_OLD, GSA_WINO_CHUNK="0"), dict(_OLD, GSA_POST_RPT="1"),
    dict(_OLD, GSA_POST_RPT="8"), {"GSA_SIDE_LEVELS": "0"}, {"GSA_MAPFUSE": "0"}, {"GSA_KSPLIT_PS": "1"}, {"GSA_FEWROWS": "0"},
    dict(_OLD, GSA_WRES="0", GSA_SUBRES="0", GSA_STATS_DIRECT="0"), dict(_OLD, GSA_WINO_GW="1"), dict(_OLD, GSA_WINO_GW="2"), dict(_OLD, GSA_WINO_PERS="32"),
    dict(_OLD, GSA_POST_NG="4"), dict(_OLD, GSA_SUBWST="0"), {"GSA_SUBWST": "0"}, {"GSA_FUSEFIN": "0"},
    # round 4's measured-slower kernels: in the experiments build only (conv3x3_wino_dma: every operand by LDS-DMA; two tiles per 8-wave workgroup)
    dict(_EXP, GSA_WINO_DMA="1"), dict(_EXP, GSA_WINO_TW="2"), dict(_EXP, GSA_SUB_CNT="1"),
    dict(_EXP, GSA_WINO_NT2="2"), dict(_EXP, GSA_WINO_NT2="1", GSA_WINO_IL="1")])
def test_speed_switches_do_not_change_the_bits(torch_cuda, tmp_path, env):
    """The A/B switches of DESIGN.md section 4 that are NOT part of the canonical arithmetic (channel tile of the Winograd
    kernel, group order, rows per thread of the post kernel, stream overlap, resident weights / persistent forms / direct
    statistics) select other kernels or launch shapes, never other results: ffhq 1024^2 batch 4 equals the oracle's digest
    under each of them.  (Child processes: the switches are read once per process.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if "GSA_HIP_LIBRARY" in env:
        from gan_segmentation_amd import _lib
        if not os.path.exists(_lib.EXPERIMENTS_LIBRARY):
            pytest.skip("libgsa_hip_exp.so is not built (make -C gan-segmentation_amd/csrc experiments)")
    script = tmp_path / "switch_worker.py"
    script.write_text(_SWITCH_WORKER.replace("ROOT_DIR", repr(root)))
    out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SWITCH_OK" in out.stdout, "%r: %s" % (env, out.stdout[-800:] + out.stderr[-2500:])
