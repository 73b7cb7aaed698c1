"""C-ABI surface (no compute without a GPU), host logic, and the sharded N>1 path on gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from gan_segmentation_amd import _lib
from gan_segmentation_amd import dist as gdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_library():
    if not os.path.exists(_lib.HIP_LIBRARY):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    return _lib.HIP_LIBRARY


def test_header_symbols_are_exported(hip_library):
    """Every function include/gsa.h declares is exported by the built library, and vice versa."""
    with open(os.path.join(ROOT, "include", "gsa.h")) as f:
        header = f.read()
    declared = set(re.findall(r"\b(gsa_[a-z_]+)\s*\(", header)) - {"gsa_ctx"}
    lib = ctypes.CDLL(hip_library)
    for name in declared:
        assert hasattr(lib, name), "%s declared in gsa.h but not exported" % name
    assert declared == {"gsa_" + s for s in _lib.API_SYMBOLS}
    api = _lib.Api(hip_library, "gsa_")
    assert b"gfx950" in api.version()


def test_training_header_symbols_are_exported(hip_library):
    """include/gsa_train.h (decoder-training operators) <-> library exports <-> the ctypes table of train_ops."""
    with open(os.path.join(ROOT, "include", "gsa_train.h")) as f:
        header = f.read()
    declared = set(re.findall(r"\bint\s+(gsa_train_[a-z0-9_]+)\s*\(", header))
    assert len(declared) == 13
    lib = ctypes.CDLL(hip_library)
    for name in declared:
        assert hasattr(lib, name), "%s declared in gsa_train.h but not exported" % name
    from gan_segmentation_amd import train_ops
    assert set(train_ops._api().keys()) == declared


def test_product_path_fails_loudly_without_gpu(hip_library):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    api = _lib.Api(hip_library, "gsa_")
    with pytest.raises(_lib.GsaError, match="no HIP device|device"):
        _lib.Context(api, 0)
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.networks_stylegan import Generator
    with pytest.raises(_lib.GsaError):
        Generator(W.reduced_generator_config(7))


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(_lib.GsaError, match="no CPU fallback"):
        _lib.Api(str(tmp_path / "libnope.so"))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan-segmentation_amd")
    for dirpath, _d, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
                assert "gsao_" not in src, fn


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            spans = [gdist.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_pack_unpack_pairs():
    import torch
    img = torch.randint(0, 255, (3, 8, 8, 3), dtype=torch.uint8)
    mask = torch.randint(0, 2, (3, 8, 8), dtype=torch.uint8)
    buf = gdist.pack_pairs(img, mask)
    assert buf.shape == (3, 8 * 8 * 4)
    i2, m2 = gdist.unpack_pairs(buf, 8)
    assert torch.equal(i2, img) and torch.equal(m2, mask)


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
from gan_segmentation_amd import dist as gdist
from gan_segmentation_amd import weights as W
from oracle.binding import Oracle
rank, world, _ = gdist.init_from_env(backend="gloo")
gcfg = W.reduced_generator_config(5)
gp = W.synthetic_generator_params(gcfg)
dcfg = W.decoder_config(5, in_channels=W.generator_channels(gcfg))
dp = W.synthetic_decoder_params(dcfg)
total = 5                                   # ragged: ranks get 3 and 2 samples
z, noise = W.synthetic_inputs(gcfg, total)
lo, hi = gdist.shard_bounds(total, world, rank)
o = Oracle(gcfg, gp, dcfg, dp)              # stands in for the HIP producer on CPU
img, mask = o.generate(z[lo:hi], [a[lo:hi] for a in noise])
counts = [gdist.shard_bounds(total, world, r)[1] - gdist.shard_bounds(total, world, r)[0] for r in range(world)]
gi, gm = gdist.gather_pairs(torch.from_numpy(img), torch.from_numpy(mask), counts=counts)
if rank == 0:
    fi, fm = o.generate(z, noise)
    assert gi.shape[0] == total
    assert np.array_equal(gi.numpy(), fi) and np.array_equal(gm.numpy(), fm)
    print("GATHER_OK")
else:
    assert gi is None
# the overlapped, copy-free gatherer of bench.py: 3 batches of 2 samples per rank through 2 slots
z6, noise6 = W.synthetic_inputs(gcfg, 12, seed_z=7, seed_noise=8)
R = 2 ** 5
gat = gdist.PairGatherer(2, R, 3, device="cpu", dst=0, depth=2)
seen = []
for k in range(3):
    slot = k & 1
    gat.wait(slot)
    lo = (k * world + rank) * 2
    bi, bm = o.generate(z6[lo:lo + 2], [a[lo:lo + 2] for a in noise6])
    img_v, mask_v = gat.buffers(slot)
    img_v.copy_(torch.from_numpy(bi)); mask_v.copy_(torch.from_numpy(bm))
    gat.submit(slot)
    gat.wait(slot)                                # CPU tensors: read back right away
    if rank == 0:
        parts = gat.result(slot)
        assert len(parts) == world
        seen.append((torch.cat([p[0] for p in parts]).numpy().copy(), torch.cat([p[1] for p in parts]).numpy().copy()))
    else:
        assert gat.result(slot) is None
gat.wait_all()
if rank == 0:
    for k, (gi2, gm2) in enumerate(seen):
        lo = k * world * 2
        fi, fm = o.generate(z6[lo:lo + 4], [a[lo:lo + 4] for a in noise6])
        assert np.array_equal(gi2, fi) and np.array_equal(gm2, fm)
    print("GATHERER_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_world_size_2_gather_on_gloo(tmp_path, oracle_lib):
    """The N>1 path: contiguous shards + one gather reproduce the single-process result."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, OMP_NUM_THREADS="2")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
        capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GATHER_OK" in out.stdout and "GATHERER_OK" in out.stdout


def test_dataset_writer_files_and_throughput(tmp_path):
    """The asynchronous writer: names, formats and contents of reference main.py:100-103."""
    import time
    from PIL import Image
    from gan_segmentation_amd.dataset_writer import DatasetWriter
    rng = np.random.default_rng(0)
    n, R = 12, 256
    ramp = np.arange(R, dtype=np.uint8)
    img = np.stack([np.broadcast_to(ramp[None, None, :], (n, R, R)), np.broadcast_to(ramp[None, :, None], (n, R, R)),
                    np.full((n, R, R), 90, np.uint8)], axis=-1).copy()    # smooth ramps survive JPEG
    mask = (rng.random((n, R, R)) > 0.5).astype(np.uint8)
    t0 = time.perf_counter()
    with DatasetWriter(str(tmp_path), workers=4) as w:
        w.submit(img[:8], mask[:8], 0)
        w.submit(img[8:], mask[8:], 8)
    assert w.written == n and time.perf_counter() - t0 < 60
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["img_%06d.jpg" % i for i in range(n)] + ["mask_%06d.png" % i for i in range(n)])
    m = np.asarray(Image.open(tmp_path / "mask_000009.png"))
    assert m.dtype == np.uint8 and np.array_equal(m, mask[9])             # PNG is lossless: class indices intact
    j = np.asarray(Image.open(tmp_path / "img_000003.jpg"))
    assert j.shape == (R, R, 3) and np.abs(j.astype(int) - img[3].astype(int)).mean() < 3



def test_dataset_writer_drain_surfaces_a_batch_level_error(tmp_path):
    """ONE batch-level failure stands for N pairs that are never counted: drain() must raise that exception at once instead
    of spinning until its timeout (bench.py's to-disk measurement would hang for ten minutes)."""
    import time
    from gan_segmentation_amd.dataset_writer import DatasetWriter
    img = np.zeros((4, 16, 16, 3), np.uint8)
    mask = np.zeros((4, 16, 16), np.uint8)
    w = DatasetWriter(str(tmp_path), workers=2)

    def boom(*a, **k):
        raise RuntimeError("the pool is gone")
    w.pool.submit = boom
    w.submit(img, mask, 0)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="the pool is gone"):
        w.drain(timeout=60.0)
    assert time.perf_counter() - t0 < 10.0
    with pytest.raises(RuntimeError, match="the pool is gone"):
        w.close()


def test_dataset_writer_withholds_the_files_of_a_batch_whose_device_check_failed(tmp_path):
    """The in-flight device checks (include/gsa.h gsa_status_snapshot, reference main.py:93-104) on the host side, without a GPU:
    the dispatcher looks at a batch's status words before it writes the batch's files; a non-zero word withholds that batch AND
    every later one, and the error names the first withheld global index."""
    import torch
    from gan_segmentation_amd.dataset_writer import DatasetWriter, DeviceCheckFailed
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (2, 32, 32, 3), dtype=np.uint8)
    mask = g.integers(0, 2, (2, 32, 32), dtype=np.uint8)
    w = DatasetWriter(str(tmp_path), workers=2)
    w.submit(img, mask, 0, status=[torch.zeros(2, dtype=torch.int32)])
    w.submit(img, mask, 2, status=[torch.zeros(2, dtype=torch.int32), torch.zeros(2, dtype=torch.int32)])      # two replicas, both clean
    assert w.drain() == 4
    w.submit(img, mask, 4, status=[torch.tensor([0, 0], dtype=torch.int32), torch.tensor([1, 0], dtype=torch.int32)])   # one replica's statistics word set
    w.submit(img, mask, 6, status=[torch.zeros(2, dtype=torch.int32)])                                         # later batch: halted too
    with pytest.raises(DeviceCheckFailed) as e:
        w.close()
    assert e.value.first_index == 4 and "global sample index 4" in str(e.value) and "instance-norm statistic" in str(e.value)
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["img_%06d.jpg" % i for i in range(4)] + ["mask_%06d.png" % i for i in range(4)])
    # the mapping time-out word is named as such
    assert "mapping network timed out" in str(DeviceCheckFailed(8, (0, 1)))


def test_mask_png_bytes_decodes_back():
    """The writer's own PNG container (zlib level 1, run-length strategy): lossless, 8-bit greyscale, valid CRCs."""
    import io
    from PIL import Image
    from gan_segmentation_amd.dataset_writer import mask_png_bytes
    rng = np.random.default_rng(1)
    for shape, k in (((16, 16), 2), ((5, 7), 8), ((256, 128), 3), ((1, 1), 2)):
        m = rng.integers(0, k, shape, dtype=np.uint8)
        im = Image.open(io.BytesIO(mask_png_bytes(m)))
        im.load()                                           # raises on a bad CRC / Adler checksum
        assert im.mode == "L" and np.array_equal(np.asarray(im), m)
    blob = np.zeros((512, 512), np.uint8)
    blob[100:400, 150:300] = 1
    assert len(mask_png_bytes(blob)) < 4096                 # constant runs collapse
    with pytest.raises(ValueError):
        mask_png_bytes(np.zeros((4, 4, 1), np.uint8))


def test_annotation_sample_files(tmp_path):
    """img_%06d.jpg + feat_%06d.pickle as the annotator saves them, mask thresholds as the
    few-shot dataset applies them (reference seg_annotator.py:322-337, seg_datasets.py:60-106)."""
    import pickle
    from PIL import Image
    from gan_segmentation_amd import annotation_io as A
    rng = np.random.default_rng(1)
    feats = [rng.normal(size=(c, r, r)).astype(np.float32) for c, r in ((32, 4), (32, 8), (16, 16))]
    img = np.full((16, 16, 3), 120, np.uint8)
    A.export_sample(str(tmp_path), 7, img, feats)
    assert sorted(os.listdir(tmp_path)) == ["feat_000007.pickle", "img_000007.jpg"]
    with open(tmp_path / "feat_000007.pickle", "rb") as fp:
        back = pickle.load(fp)                                   # a plain list of CHW fp32 arrays
    assert isinstance(back, list) and all(np.array_equal(a, b) and a.dtype == np.float32 for a, b in zip(back, feats))
    Image.fromarray(np.array([[0, 63, 64], [192, 193, 255]], np.uint8), "L").save(tmp_path / "mask_000007.png")
    Image.fromarray(np.full((2, 3, 3), 10, np.uint8), "RGB").save(tmp_path / "img_000007.jpg")
    mask, img2, f2 = A.load_sample(str(tmp_path), 7)
    assert mask.tolist() == [[-1, -1, 0], [0, 1, 1]] and img2.shape == (2, 3, 3) and len(f2) == 3


def test_bench_reads_every_committed_pmc_summary():
    """bench.py takes `roofline.traffic` from profiles/*_pmc_summary.json: every committed summary must be in the
    layout it reads (a list of records keyed "kernel"), and an unknown kernel gives None, never an exception."""
    import glob
    import json
    import bench
    paths = glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))
    assert paths
    for path in paths:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
        assert isinstance(kernels, list) and all("kernel" in k and "hbm_bytes_per_launch" in k for k in kernels), path
        hit = bench.pmc_traffic(kernels[0]["kernel"])
        assert hit is not None and hit["bytes_per_launch"] > 0
    assert bench.pmc_traffic("no such kernel") is None


_WORKER4 = r'''
import os, sys
sys.path.insert(0, ROOT_DIR)
import numpy as np, torch
import torch.distributed as dist
from gan_segmentation_amd import dist as gdist
from gan_segmentation_amd import main as cli
rank, world, _ = gdist.init_from_env(backend="gloo")
assert world == 4
R, ch, B = 8, 3, 3

def pair(index):        # a recognisable (image, mask) per GLOBAL sample index -- stands in for the HIP producer
    g = np.random.default_rng(index)
    return g.integers(0, 256, (R, R, ch), dtype=np.uint8), g.integers(0, 2, (R, R), dtype=np.uint8)

# ---- (1) PairGatherer, depth 2, five batches: every slot is reused (wrap-around) while the previous gather of
# that slot is awaited only when the slot comes round again, as bench.py does
gat = gdist.PairGatherer(B, R, ch, device="cpu", dst=0, depth=2)
for k in range(5):
    slot = k % 2
    gat.wait(slot)
    if rank == 0 and k >= 2:      # what the slot held from batch k-2 must have been complete before it is overwritten
        parts = gat.result(slot)
        for r in range(world):
            for j in range(B):
                wi, wm = pair(((k - 2) * world + r) * B + j)
                assert np.array_equal(parts[r][0][j].numpy(), wi) and np.array_equal(parts[r][1][j].numpy(), wm), (k, r, j)
    iv, mv = gat.buffers(slot)
    for j in range(B):
        wi, wm = pair((k * world + rank) * B + j)
        iv[j].copy_(torch.from_numpy(wi)); mv[j].copy_(torch.from_numpy(wm))
    gat.submit(slot)
gat.wait_all()
if rank == 0:
    for k in (3, 4):
        parts = gat.result(k % 2)
        for r in range(world):
            for j in range(B):
                wi, wm = pair((k * world + r) * B + j)
                assert np.array_equal(parts[r][0][j].numpy(), wi) and np.array_equal(parts[r][1][j].numpy(), wm)
    print("GATHERER4_OK")

# ---- (2) ragged last batch through the blocking form: 10 samples over 4 ranks = 3, 3, 2, 2
total = 10
lo, hi = gdist.shard_bounds(total, world, rank)
counts = [gdist.shard_bounds(total, world, r)[1] - gdist.shard_bounds(total, world, r)[0] for r in range(world)]
assert counts == [3, 3, 2, 2]
mine = [pair(1000 + i) for i in range(lo, hi)]
gi, gm = gdist.gather_pairs(torch.from_numpy(np.stack([p[0] for p in mine])), torch.from_numpy(np.stack([p[1] for p in mine])), counts=counts)
if rank == 0:
    assert gi.shape[0] == total
    for i in range(total):
        wi, wm = pair(1000 + i)
        assert np.array_equal(gi[i].numpy(), wi) and np.array_equal(gm[i].numpy(), wm)
    print("RAGGED4_OK")
else:
    assert gi is None and gm is None

# ---- (3) `main.py generate` under torchrun: the file indices of the ranks partition [0, GENERATE_NUM)
for n_generate, batch in ((10, 4), (3, 8), (10000, 32), (17, 1)):
    mine = [i for first, bs in cli.shard_batches(n_generate, batch, world, rank) for i in range(first, first + bs)]
    sizes = [bs for _f, bs in cli.shard_batches(n_generate, batch, world, rank)]
    assert all(0 < b <= batch for b in sizes) and all(b == batch for b in sizes[:-1])
    every = [None] * world
    dist.all_gather_object(every, mine)
    if rank == 0:
        flat = [i for part in every for i in part]
        assert flat == list(range(n_generate)), (n_generate, batch)          # disjoint, complete, in rank order
r2, w2, l2 = gdist.env_ranks()
assert (r2, w2) == (rank, world)
assert cli.devices_for_rank({"GAN_GPU_IDS": [4, 5]}, world, l2) == ([[4, 5][l2 % 2]],) * 2
assert cli.devices_for_rank({"GAN_GPU_IDS": []}, world, l2) == ([l2], [l2])
if rank == 0:
    print("SHARDS4_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_world_size_4_gatherer_and_generate_shards_on_gloo(tmp_path):
    """Multi-GPU readiness without hardware: four ranks over gloo -- the overlapped gatherer with slot wrap-around, the
    ragged blocking gather, and the shard arithmetic of `main.py generate` (reference image_generator.py:95:
    split_and_load(even_split=False); main.py:93-103)."""
    script = tmp_path / "worker4.py"
    script.write_text(_WORKER4.replace("ROOT_DIR", repr(ROOT)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
         "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
        capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    for tag in ("GATHERER4_OK", "RAGGED4_OK", "SHARDS4_OK"):
        assert tag in out.stdout, out.stdout[-2000:]


_WORKER_BENCH = r"""
import os, sys
sys.path.insert(0, ROOT_DIR)
import numpy as np, torch
import torch.distributed as dist
from gan_segmentation_amd import dist as gdist
import bench
rank, world, _ = gdist.init_from_env(backend="gloo")
assert world == 2
R, ch, B = 8, 3, 2

def pair(r, step, j):      # recognisable bytes per (rank, step, sample): stands in for gsa_generate
    g = np.random.default_rng(1000 * r + 10 * step + j)
    return g.integers(0, 256, (R, R, ch), dtype=np.uint8), g.integers(0, 2, (R, R), dtype=np.uint8)

class Stub:
    def __init__(self): self.calls = 0
    def __call__(self, out):
        step = self.calls; self.calls += 1
        imgs = torch.from_numpy(np.stack([pair(rank, step, j)[0] for j in range(B)]))
        masks = torch.from_numpy(np.stack([pair(rank, step, j)[1] for j in range(B)]))
        if out is None:
            return imgs, masks
        out[0].copy_(imgs); out[1].copy_(masks)
        return out

class Refusing(gdist.PairGatherer):       # an RCCL build without async gather into views
    def __init__(self, *a, who=(0, 1), **k):
        super().__init__(*a, **k); self.who = who
    def submit(self, slot):
        if self.rank in self.who: raise RuntimeError("async gather refused")
        return super().submit(slot)

# ---- (1) the overlapped path: 2 warm-up + 3 timed steps, content and order of the last batch on rank 0
stub = Stub()
gat = gdist.PairGatherer(B, R, ch, device="cpu", dst=0, depth=2)
loop = bench.TimedLoop(stub, gat, world, "cpu", allow_blocking=False)
t = loop.run(2, 3)
assert stub.calls == 5 and t["gather"] == "overlapped" and t["last_slot"] == 0 and t["dt_local"] > 0
dt = loop.max_over_ranks(float(rank + 1))
assert dt == 2.0                                                    # MAX over ranks
if rank == 0:
    parts = gat.result(t["last_slot"])
    assert len(parts) == world
    for r in range(world):
        for j in range(B):
            wi, wm = pair(r, 4, j)
            assert np.array_equal(parts[r][0][j].numpy(), wi) and np.array_equal(parts[r][1][j].numpy(), wm), (r, j)
    print("BENCH_OVERLAPPED_OK")
else:
    assert gat.result(t["last_slot"]) is None
dist.barrier()

# ---- (2) refusal without --allow-blocking: exit code 3 on the refusing ranks (every rank here)
loop = bench.TimedLoop(Stub(), Refusing(B, R, ch, device="cpu", dst=0, depth=2), world, "cpu", allow_blocking=False)
try:
    loop.run(1, 1)
    raise AssertionError("refused gather did not stop the run")
except SystemExit as e:
    assert e.code == 3
dist.barrier()

# ---- (3) --allow-blocking, every rank refused: all fall back in the first warm-up step, collectives still match
loop = bench.TimedLoop(Stub(), Refusing(B, R, ch, device="cpu", dst=0, depth=2), world, "cpu", allow_blocking=True)
t = loop.run(2, 2)
assert t["gather"] == "blocking" and loop.gather == "blocking"
dist.barrier()

# ---- (4) --allow-blocking, ONLY rank 1 refused: rank 0 learns of it from the MAX all-reduce after warm-up
loop = bench.TimedLoop(Stub(), Refusing(B, R, ch, device="cpu", dst=0, depth=2, who=(1,)), world, "cpu", allow_blocking=True)
for _ in range(2):
    loop.step()
loop.fence()
assert loop.gather == ("blocking" if rank == 1 else "overlapped")
loop.agree_on_fallback()
assert loop.gather == "blocking"                                   # agreed across ranks
for _ in range(2):
    loop.step()
loop.fence()
if rank == 0:
    print("BENCH_FALLBACK_OK")
dist.barrier()
dist.destroy_process_group()
"""


def test_bench_timed_loop_world_size_2_on_gloo(tmp_path):
    """bench.py's N>1 control flow (step / fence / fallback flag / MAX reduce) executed on two gloo ranks with a stub
    producer: the driver's 8-GPU run is then not the first execution of that code (reference image_generator.py:95-114)."""
    script = tmp_path / "worker_bench.py"
    script.write_text(_WORKER_BENCH.replace("ROOT_DIR", repr(ROOT)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", "29623", str(script)],
        capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    for tag in ("BENCH_OVERLAPPED_OK", "BENCH_FALLBACK_OK"):
        assert tag in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def _bench_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    import json
    return json.loads(lines[0])


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the shape of the driver's N=1 command): the parent starts two
    fresh rank processes, the real main() runs on gloo with the stub producer, ONE line comes back and it proves its ranks
    (reference image_generator.py:95-114: the batch split over the device list + gather)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-producer", "--steps", "3",
                          "--warmup", "2", "--batch", "2"], capture_output=True, text=True, timeout=300, env=_bench_env())
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = _one_json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and len(line["devices"]) == 2
    assert line["config"]["gather"] == "overlapped" and line["config"]["backend"] == "gloo"
    assert line["launcher"].startswith("self-launched") and line["data"] == "stub"
    assert line["gathered_matches_producers"] is True and line["steps"] == 3 and line["config"]["global_batch"] == 4


def test_bench_under_torchrun_is_unchanged_and_a_dead_rank_fails_the_run():
    """Under torch.distributed.run (the documented N>1 command) nothing is re-launched; and without a launcher a rank that
    exits non-zero (here: no HIP device for the real producer) makes the parent exit non-zero instead of hanging."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-producer",
                          "--steps", "2", "--warmup", "1", "--batch", "2"], capture_output=True, text=True, timeout=300, env=_bench_env())
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = _one_json_line(out.stdout)
    assert line["ranks"] == 2 and line["launcher"].startswith("external") and line["config"]["gather"] == "overlapped"
    import torch
    if not torch.cuda.is_available():
        bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                             capture_output=True, text=True, timeout=300, env=_bench_env())
        assert bad.returncode != 0 and "{" not in bad.stdout, bad.stdout[-1000:]
        assert "stopping the other ranks" in bad.stderr or "needs a HIP device" in bad.stderr


def test_bench_roofline_fields_are_fractions():
    """`roofline.frac` is what the pipe does: executed FLOP / time / peak for an MFMA-bound kernel, algorithmic bytes / time /
    8 TB/s for an HBM-bound one -- never the reference-formulation throughput, which moved to algorithmic_*."""
    import bench
    wino = {"name": "k", "ms": 1.0495, "launches": 5, "flops": 5 * 14.6e9, "alg_flops": 5 * 32.86e9, "bytes": 5 * 103e6}
    r = bench.roofline_of(wino, "fp32", None)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.43 < r["frac"] < 0.45 and r["achieved"] == r["executed_tflops"]
    assert r["algorithmic_frac"] > 0.99 and r["traffic"] is None
    blur = {"name": "post", "ms": 0.27, "launches": 1, "flops": 0.0, "alg_flops": 0.0, "bytes": 1.07e9}
    r = bench.roofline_of(blur, "fp32", {"bytes_per_launch": 1.2e9, "source": "x.json"})
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0.45 < r["frac"] < 0.55 and r["traffic"] == 1.2e9
    t_mfma, t_hbm = bench.roof_seconds([wino, blur], "fp32")
    assert abs(t_mfma - 5 * 14.6e9 / 157.3e12) < 1e-9 and abs(t_hbm - 1.07e9 / 6.29e12) < 1e-9
    # bf16 mode: the same fields against the dense bf16 MFMA peak -- the r04 cars line carried 2.052 here because the
    # f32 peak was used whatever the precision.  The dominant kernel of that run: 22.6 GFLOP executed and algorithmic, 0.07 ms, 0.19 GB
    conv = {"name": "c", "ms": 0.070, "launches": 1, "flops": 22.6e9, "alg_flops": 22.6e9, "bytes": 0.19e9}
    r = bench.roofline_of(conv, "bf16", None)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.3 < r["frac"] < 0.4 and r["mfma_peak_tflops"] == 2500.0
    for key in ("frac", "executed_frac_of_mfma_peak", "algorithmic_frac", "hbm_frac"):
        assert 0.0 <= r[key] <= 1.0, (key, r[key])
    assert abs(r["executed_frac_of_mfma_peak"] - 22.6e9 / 0.070e-3 / 2500e12) < 1e-3
    t_mfma, t_hbm = bench.roof_seconds([conv], "bf16")
    assert t_mfma == 0.0 and abs(t_hbm - 0.19e9 / 6.29e12) < 1e-12


def test_bench_timed_loop_repeats_and_median():
    """one warm-up, then R regions of exactly K steps each between two fences; dt_local is the median region."""
    import bench

    class Gat:
        depth = 2
        def wait(self, slot): pass
        def buffers(self, slot): return (None, None)
        def submit(self, slot): pass
        def wait_all(self): pass
    calls = []
    loop = bench.TimedLoop(lambda out: calls.append(1), Gat(), 1, "cpu", False, sync=lambda: calls.append("fence"))
    t = loop.run(2, 3, 5)
    assert calls.count(1) == 2 + 3 * 5 and calls.count("fence") == 1 + 5
    assert len(t["dts_local"]) == 5 and t["dt_local"] == sorted(t["dts_local"])[2]
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0]) == 4.0


def test_generate_devices_single_process():
    """One process (the reference's way): the whole GAN_GPU_IDS list is used in-process, the batch is
    GAN_BATCH_SIZE_PER_GPU * len(GAN_GPU_IDS) (reference main.py:87-88); an empty list has no CPU fallback."""
    from gan_segmentation_amd import main as cli
    assert cli.devices_for_rank({"GAN_GPU_IDS": [0, 1, 2], "SOLVER_GPU_IDS": [1]}, 1, 0) == ([0, 1, 2], [1])
    assert cli.devices_for_rank({"GAN_GPU_IDS": [3]}, 1, 0) == ([3], [3])
    with pytest.raises(RuntimeError):
        cli.devices_for_rank({"GAN_GPU_IDS": []}, 1, 0)
    assert [b for b in cli.shard_batches(5, 2, 1, 0)] == [(0, 2), (2, 2), (4, 1)]
