#!/usr/bin/env python3
"""Headline benchmark: synthetic (image, mask) pairs/sec of the `generate` hot path --
FFHQ-1024 StyleGAN synthesis + 2-class decoder, fp32, batch 8 per GPU (BASELINE.json
configs[1]); N GPUs = N processes, each its own batch (weak scaling), ONE RCCL gather of the
uint8 pairs to rank 0 per step.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the bench contract in DESIGN.md)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_SAMPLE = {"ffhq": 126.44, "cars": 85.52, "bedrooms": 55.51}
MB_PER_SAMPLE = {"ffhq": 1440.0, "cars": 623.2, "bedrooms": 234.4}
PEAK_FP32_TFLOPS = 157.3    # MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA (the headline figure with 2:1 sparsity is not used)
PEAK_HBM_GBS = 8000.0
MEASURED_HBM_GBS = 6290.0   # copy bandwidth this part reaches (MI355X_MICROARCH.md, HBM section)


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes
    (profiles/*_pmc_summary.json, produced by tools/pmc.sh + tools/pmc_report.py: separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if absent."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            with open(path) as f:
                for k in json.load(f)["kernels"]:
                    if k["kernel"] == kernel_name:
                        best = {"bytes_per_launch": round(k["hbm_bytes_per_launch"]), "source": os.path.basename(path)}
        except (OSError, ValueError, KeyError, TypeError):      # a summary in another layout must never break the bench
            pass
    return best


def host_cores():
    """Threads this process may really use: its CPU affinity capped by the cgroup CPU quota -- every CPU it is allowed,
    no cap of our own (`GSA_CPU_THREADS` overrides)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                quota = max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    quota = max(1, q // p)
        except (OSError, ValueError):
            pass
    if os.environ.get("GSA_CPU_THREADS"):
        return int(os.environ["GSA_CPU_THREADS"])
    if quota is not None:
        return max(1, min(n, quota))
    return n


def cpu_baseline(gan, seconds_budget=25.0):
    """Stand-in for the reference's mxnet-CPU path (MXNet is not installable here): the semantic torch-CPU restatement
    (oneDNN convolutions), batch 1, on EVERY CPU this process may use; where that is more than 16 threads a short second leg
    with 16 is timed too (a batch-1 convolution does not always scale to a whole host) and the FASTER of the two is `value` --
    the choice that flatters the GPU/CPU ratio least."""
    import torch
    from gan_segmentation_amd import weights as W
    from oracle import ref_semantic as S
    mr = W.GAN_MAX_RES_LOG2[gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)

    def leg(threads, budget, most):
        torch.set_num_threads(threads)
        times, i, t_start = [], 0, time.perf_counter()
        while True:
            z, noise = W.synthetic_inputs(gcfg, 1, seed_z=100 + i, seed_noise=200 + i)
            t0 = time.perf_counter()
            S.generate(gcfg, gp, dcfg, dp, z, noise)
            dt = time.perf_counter() - t0
            if i > 0:       # first sample is the warm-up
                times.append(dt)
            i += 1
            if (len(times) >= 1 and time.perf_counter() - t_start > budget) or len(times) >= most:
                break
        times.sort()
        return 1.0 / times[len(times) // 2], len(times), torch.get_num_threads()

    cores = host_cores()
    legs = [leg(cores, seconds_budget * (0.6 if cores > 16 else 1.0), 16)]
    if cores > 16:
        legs.append(leg(16, seconds_budget * 0.4, 8))
    best = max(legs, key=lambda l: l[0])
    return {"value": best[0], "unit": "pairs/s", "cores": best[2], "host_cpus": os.cpu_count(), "usable_cpus": cores, "kind": "port",
            "legs": [{"threads": l[2], "pairs_per_s": round(l[0], 4), "samples": l[1]} for l in legs],
            "sample": "%s %d^2 synthesis+decoder, batch 1, %d samples after 1 warm-up, median; torch-CPU fp32 "
                      "(oneDNN) restatement oracle/ref_semantic.py standing in for the reference's mxnet-CPU path"
                      % (gan, 2 ** mr, best[1])}


def pair_digest(img, mask):
    """SHA-256 of one sample's (image, mask) bytes -- the digest tests/golden/bench_outputs.json holds for the C
    oracle's output on these very inputs (tests/golden/make_bench_hash.py)."""
    import hashlib
    return hashlib.sha256(img.contiguous().cpu().numpy().tobytes() + mask.contiguous().cpu().numpy().tobytes()).hexdigest()


def golden_digests(gan, batch):
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_outputs.json")) as f:
            return json.load(f).get("%s_b%d" % (gan, batch), {}).get("samples")
    except (OSError, ValueError):
        return None


def output_check(gan, batch, precision, img, mask):
    """What the run produced, so that the line is not only a speed: digest of sample 0 and whether it equals the
    oracle's (fp32, the batch sizes the golden file holds; bf16 mode is a stated tolerance, never bit equality)."""
    out = {"sample0_sha256": pair_digest(img[0], mask[0]), "mask_mean": round(float(mask.float().mean().item()), 6)}
    want = golden_digests(gan, batch) if precision == "fp32" else None
    out["oracle_sha256"] = want[0] if want else None
    out["matches_oracle"] = (out["sample0_sha256"] == want[0]) if want else None
    return out


def measure_secondary(gan, batch, precision, steps, warmup, dev, graph=False):
    """A short, separate measurement of another BASELINE.json configuration (own model, own context), reported under
    `secondary` -- the headline fields stay those of configs[1].  graph=True: the step is captured into a hipGraph during
    the warm-up (what a steady `generate` loop reaches after 32 identical calls) and the timed steps replay it; the entry
    says which form was timed."""
    import torch
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.image_generator import ImageGenerator
    mr = W.GAN_MAX_RES_LOG2[gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[dev.index], batch_size=batch, precision=precision)
    gen.graph_mode = "1" if graph else "0"
    gen.graph_after = 3
    z, noise = W.synthetic_inputs(gcfg, batch, seed_z=1000, seed_noise=2000)
    z = torch.from_numpy(z).to(dev)
    noise = [torch.from_numpy(a).to(dev) for a in noise]
    R = 2 ** mr
    out = (torch.empty((batch, R, R, gcfg["channels"]), device=dev, dtype=torch.uint8),
           torch.empty((batch, R, R), device=dev, dtype=torch.uint8))      # fixed addresses: the key of the captured graph
    for _ in range(warmup):
        img, mask = gen.generate_batch(z, noise, out=out)
    torch.cuda.synchronize()
    captured = gen.graphs_captured()
    t0 = time.perf_counter()
    for _ in range(steps):
        img, mask = gen.generate_batch(z, noise, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"workload": "stylegan-%s %d^2 synthesis + decoder, batch=%d, %s, 1 GPU" % (gan, 2 ** mr, batch, precision),
           "value": round(batch * steps / dt, 2), "unit": "pairs/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * dt / steps, 3), "dtype": "f32" if precision == "fp32" else "bf16",
           "graph": "replayed (captured in the warm-up)" if captured and gen.graphs_captured() == captured else "eager",
           "output": output_check(gan, batch, precision, img, mask)}
    del gen
    return res


def measure_end_to_end(gen, B, steps, dev):
    """The dataset rate beside the kernel-path rate: what `main.py generate` does per batch besides gsa_generate --
    counter-based latents/noise drawn on the device (gsa_fill_inputs), the GPU JPEG + PNG encoders, the copy of the
    compressed bytes to the host and the file writes (DatasetWriter into a temporary directory)."""
    import shutil
    import tempfile
    import torch
    from gan_segmentation_amd.dataset_writer import DatasetWriter
    out = {}
    for _ in range(2):
        gen.generate_indexed(0, B, seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        gen.generate_indexed(k * B, B, seed=0)
    torch.cuda.synchronize()
    out["with_input_generation_pairs_per_s"] = round(B * steps / (time.perf_counter() - t0), 2)
    tmp = tempfile.mkdtemp(prefix="gsa_bench_")
    try:
        # ONE writer: its encoders, pinned buffers and threads are created by the warm-up batches (one per slot and more),
        # the timed region is submit -> all files of those batches on disk (drain), steady state as in a 10 000-sample run
        with DatasetWriter(tmp, gpu_jpeg=True, gpu_png=True) as w:
            for k in range(6):
                w.submit(*gen.generate_indexed(k * B, B, seed=0), k * B)
            w.drain()
            torch.cuda.synchronize()
            nsteps = max(steps, 24)
            t0 = time.perf_counter()
            for k in range(nsteps):
                w.submit(*gen.generate_indexed((6 + k) * B, B, seed=0), (6 + k) * B)
            w.drain()
            dt = time.perf_counter() - t0
        out["to_disk_pairs_per_s"] = round(B * nsteps / dt, 2)
        out["to_disk_files"] = len(os.listdir(tmp))
        out["sink"] = ("img_%%06d.jpg + mask_%%06d.png in a temporary directory (GPU JPEG/PNG encoders, host threads frame and write); "
                       "timed: submit of %d batches until all their files are on disk" % nsteps)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def measure_box(dev):
    """Two figures that describe the BOX the line was measured on, not this repository (the pool's boxes differ by a few per cent for the
    same build: `box` lets two lines be compared): what a 512 MiB device-to-device copy reaches (torch's copy kernel; read + written bytes)
    and what an fp32 4096^3 GEMM of the vendor library reaches (the matrix pipe at whatever clock the box sustains)."""
    import torch
    n = 128 * 1024 * 1024
    x = torch.ones(n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    a = torch.randn(4096, 4096, dtype=torch.float32, device=dev)
    b = torch.randn(4096, 4096, dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize(dev)
        best = float("inf")
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            best = min(best, e0.elapsed_time(e1) / reps)
        return best

    copy_ms = timed(lambda: y.copy_(x), 10)
    gemm_ms = timed(lambda: torch.mm(a, b), 5)
    return {"copy_512MiB_tb_per_s": round(2 * n * 4 / copy_ms / 1e9, 2), "sgemm_4096_tflops": round(2 * 4096.0 ** 3 / gemm_ms / 1e9, 1),
            "note": "box calibration (torch copy kernel / vendor fp32 GEMM), independent of this repository's kernels"}


def measure_generate_job(gan, n, batch, dev):
    """BASELINE.json configs[2]'s workload on ONE card, as the reference runs it (main.py:75-104): `main.py generate` end to end --
    .params files loaded from disk, `n` samples in batches of `batch` (the 8-GPU node's global batch of 32 on a single GPU),
    counter-based inputs, GPU JPEG/PNG encoders, the in-flight status checks, img_%06d.jpg + mask_%06d.png on disk.  Timed: the
    whole call, model load included, after a one-batch call that only warms the process (library load, first allocation)."""
    import shutil
    import tempfile
    from gan_segmentation_amd import main as cli
    from gan_segmentation_amd import params as P
    from gan_segmentation_amd import weights as W
    mr = W.GAN_MAX_RES_LOG2[gan]
    d = tempfile.mkdtemp(prefix="gsa_job_")
    try:
        os.makedirs(os.path.join(d, "models"))
        os.makedirs(os.path.join(d, "exp", "checkpoints"))
        P.save_params(os.path.join(d, "models", "stylegan-%s.params" % gan), W.synthetic_generator_params(W.generator_config(mr)))
        P.save_params(os.path.join(d, "exp", "checkpoints", "checkpoint_last.params"), W.synthetic_decoder_params(W.decoder_config(mr)))
        cfg = {"BASE_DIR": os.path.join(d, "exp"), "GAN": gan, "GAN_DIR": os.path.join(d, "models"), "GAN_GPU_IDS": [dev.index],
               "GAN_BATCH_SIZE_PER_GPU": batch, "SOLVER_GPU_IDS": [dev.index], "ANNOTATION": "segmentation", "GENERATE_NUM": n}
        rc = cli.generate(cfg, limit=batch)
        t0 = time.perf_counter()
        rc = rc or cli.generate(cfg, limit=n)
        dt = time.perf_counter() - t0
        dst = os.path.join(d, "exp", "dataset", "train_generated")
        names = os.listdir(dst)
        return {"generate_job": "main.py generate: %s, %d samples, batch %d, 1 GPU (BASELINE.json configs[2]'s workload on one card), "
                                "synthetic .params loaded from disk, files written to a temporary directory" % (gan, n, batch),
                "generate_job_rc": rc, "generate_job_s": round(dt, 3), "generate_job_pairs_per_s": round(n / dt, 2),
                "generate_job_files": len(names), "generate_job_bytes": sum(os.path.getsize(os.path.join(dst, f)) for f in names)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


class TimedLoop:
    """The N-rank control flow of the timed region, free of any GPU call so that a world-size-2 gloo test can drive it
    with a stub producer (tests/test_abi_and_host.py): per step wait for the slot's previous gather, produce into the
    slot's send views, submit ONE asynchronous gather (reference image_generator.py:95-114: split over the ctx list +
    host gather); `fence` = every gather done + barrier + device sync; after warm-up the ranks agree (MAX all-reduce)
    on the blocking fallback, because the collectives of all ranks must match; `dt` is the MAX over ranks.

    produce(out): write one batch of pairs into out = (img view, mask view); with out=None return (img, mask)."""

    def __init__(self, produce, gat, world, device, allow_blocking, sync=None):
        self.produce, self.gat, self.world, self.device = produce, gat, world, device
        self.allow_blocking, self.sync = allow_blocking, sync
        self.k = 0
        self.gather = "overlapped" if world > 1 else "none"

    def step(self):
        from gan_segmentation_amd import dist as gdist
        slot = self.k % self.gat.depth
        self.k += 1
        if self.gather == "blocking":        # fallback (see below): the plain blocking gather
            gdist.gather_pairs(*self.produce(None))
            return
        self.gat.wait(slot)                  # the gather that last read this buffer (`depth` batches ago)
        self.produce(self.gat.buffers(slot))
        try:
            self.gat.submit(slot)
        except Exception as e:               # an RCCL build without async gather into views
            if not self.allow_blocking:
                print("bench: the overlapped gather was refused (%s); rerun with --allow-blocking to time the blocking "
                      "gather instead" % e, file=sys.stderr, flush=True)
                raise SystemExit(3)
            print("bench: overlapped gather failed (%s); using the blocking gather" % e, file=sys.stderr, flush=True)
            self.gather = "blocking"
            gdist.gather_pairs(*self.gat.buffers(slot))

    def fence(self):
        import torch.distributed as dist
        self.gat.wait_all()
        if self.world > 1:
            dist.barrier()
        if self.sync is not None:
            self.sync()

    def agree_on_fallback(self):
        """A rank that fell back during warm-up takes every rank with it (the collectives must match)."""
        import torch
        import torch.distributed as dist
        if self.world > 1:
            flag = torch.tensor([1 if self.gather == "blocking" else 0], device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                self.gather = "blocking"

    def max_over_ranks(self, seconds):
        import torch
        import torch.distributed as dist
        if self.world > 1:
            t = torch.tensor([seconds], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seconds = float(t.item())
        return seconds

    def run(self, warmup, steps, repeats=1):
        """Warm-up, then `repeats` timed regions back to back, each EXACTLY `steps` steps between two fences (barrier + device
        sync on both sides).  `dts_local` holds every region's seconds, `dt_local` their median."""
        t1 = time.perf_counter()
        for _ in range(warmup):
            self.step()
        self.fence()
        self.agree_on_fallback()
        t_warm = time.perf_counter() - t1
        dts = []
        for _ in range(max(1, repeats)):
            t0 = time.perf_counter()
            for _ in range(steps):               # EXACTLY `steps` steps between two fences
                self.step()
            self.fence()
            dts.append(time.perf_counter() - t0)
        return {"warmup_s": t_warm, "dt_local": median(dts), "dts_local": dts, "gather": self.gather,
                "last_slot": (self.k - 1) % self.gat.depth}


def median(xs):
    """Middle element of the sorted values (the upper one of an even count): always one of the measured regions."""
    ys = sorted(xs)
    return ys[len(ys) // 2]


def self_launch(n, argv):
    """`python3 bench.py --gpus N` with no launcher around it: THIS process -- which has not imported torch or touched HIP, and
    never does -- starts N fresh child interpreters of the same command line, one rank per GPU, with the rendezvous variables
    torchrun would set (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT), relays
    rank 0's single JSON line, and returns non-zero as soon as any child does (the others are then stopped by PID).  Under
    torchrun WORLD_SIZE is set and this is never reached."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GSA_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("bench: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr, flush=True)
                for q in live:
                    procs[q].terminate()          # exactly the PIDs started above
        time.sleep(0.05)
    t.join(timeout=10)
    return rc


class StubProducer:
    """`--stub-producer`: deterministic uint8 pairs instead of the generate kernels, so that the whole N-rank control flow of
    this file (self-launch, rendezvous, gatherer, TimedLoop, the line) runs on gloo without a GPU (tests/test_abi_and_host.py)."""

    def __init__(self, rank, B, R, ch):
        self.rank, self.B, self.R, self.ch, self.calls = rank, B, R, ch, 0

    def pairs(self, step):
        import numpy as np
        import torch
        g = np.random.Generator(np.random.PCG64(1000 * self.rank + step))
        return (torch.from_numpy(g.integers(0, 256, (self.B, self.R, self.R, self.ch), dtype=np.uint8)),
                torch.from_numpy(g.integers(0, 2, (self.B, self.R, self.R), dtype=np.uint8)))

    def __call__(self, out):
        img, mask = self.pairs(self.calls)
        self.calls += 1
        if out is None:
            return img, mask
        out[0].copy_(img)
        out[1].copy_(mask)
        return out


def roofline_of(top, precision, traffic):
    """The `roofline` object for the dominant kernel: `achieved` / `frac` are what the PIPE does -- the FLOP the matrix cores
    really issue / time / the f32 MFMA peak for an MFMA-bound kernel, the algorithmic bytes / time / 8 TB/s for an HBM-bound
    one -- so no field named `frac` can exceed 1.  The throughput in the reference's formulation (2*MACs of the direct
    convolution, SURVEY.md section 8d: larger than the executed count for the Winograd and sub-pixel kernels) is reported
    separately as `algorithmic_tflops` / `algorithmic_frac`, which CAN exceed 1 and is not a utilisation."""
    sec = top["ms"] * 1e-3
    launches = max(1, top["launches"])
    mfma_peak = PEAK_FP32_TFLOPS if precision == "fp32" else PEAK_BF16_TFLOPS      # the MFMA peak of the precision the kernel multiplies in
    ex_tf = top["flops"] / sec / 1e12 if sec > 0 else 0.0
    alg_tf = top["alg_flops"] / sec / 1e12 if sec > 0 else 0.0
    gbs = top["bytes"] / sec / 1e9 if sec > 0 else 0.0
    t_mfma = top["flops"] / (mfma_peak * 1e12)
    t_hbm = top["bytes"] / (PEAK_HBM_GBS * 1e9)
    hbm_bound = t_hbm > t_mfma     # bf16 mode: 16x the MFMA rate, the same kernels sit under the HBM roof
    if hbm_bound:
        bound, ach, peak, unit = "hbm", gbs, PEAK_HBM_GBS, "GB/s"
    else:
        bound, ach, peak, unit = "mfma", ex_tf, mfma_peak, "TFLOP/s"
    return {"bound": bound, "kernel": top["name"], "achieved": round(ach, 3), "peak": peak, "unit": unit,
            "frac": round(ach / peak, 4),
            "mfma_peak_tflops": mfma_peak,
            "executed_tflops": round(ex_tf, 3), "executed_frac_of_mfma_peak": round(ex_tf / mfma_peak, 4),
            "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
            "algorithmic_tflops": round(alg_tf, 3), "algorithmic_frac": round(alg_tf / mfma_peak, 4),
            "avg_launch_ms": round(top["ms"] / launches, 4), "launches": top["launches"],
            "note": "achieved/frac = EXECUTED FLOP of the kernel (Winograd F(2x2,3x3): 16 products per 4 outputs, F(2x2,2x2): 9 per 4; "
                    "sub-pixel form: 4 taps instead of 9) / time / the MFMA peak of the precision (157.3 TF f32, 2500 TF dense bf16) -- the "
                    "utilisation of the matrix pipe; "
                    "algorithmic_* = the same layers in the reference's direct formulation (SURVEY.md 8d), a throughput that can exceed the peak",
            "measured": "HIP events around every launch in a second pass of the same K steps, run right after the timed region with "
                        "the decoder-beside-synthesis stream overlap off (durations not stretched by a concurrent kernel)",
            "executed_flops_per_launch": round(top["flops"] / launches),
            "algorithmic_flops_per_launch": round(top["alg_flops"] / launches),
            "algorithmic_bytes_per_launch": round(top["bytes"] / launches),
            "traffic": traffic["bytes_per_launch"] if traffic else None,
            "traffic_source": traffic["source"] if traffic else None}


def roof_seconds(entries, precision):
    """The step's own roof: per kernel label max(executed FLOP / MFMA peak, algorithmic bytes / the 6.29 TB/s this part's copy
    kernels reach), summed -- MFMA time for the convolutions, HBM time for the byte passes (blur, toRGB, final conv, finalize)."""
    peak = (PEAK_FP32_TFLOPS if precision == "fp32" else PEAK_BF16_TFLOPS) * 1e12
    t_mfma = sum(e["flops"] / peak for e in entries if e["flops"] / peak >= e["bytes"] / (MEASURED_HBM_GBS * 1e9))
    t_hbm = sum(e["bytes"] / (MEASURED_HBM_GBS * 1e9) for e in entries if e["flops"] / peak < e["bytes"] / (MEASURED_HBM_GBS * 1e9))
    return t_mfma, t_hbm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (default 50: a third of a second of GPU time at FFHQ batch 8)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-steps", type=int, default=80,
                    help="untimed steps in front of the warm-up (default 80 = half a second at FFHQ batch 8): an idle MI355X needs a few tenths of a "
                         "second of load before its clocks settle -- without them the first one or two timed regions read 3 %% low (the same count on every rank)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each, back to back after ONE warm-up; value = their median, every region is printed (runs)")
    ap.add_argument("--gan", default="ffhq", choices=("ffhq", "cars", "bedrooms"))
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU per step")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16"),
                    help="bf16 = bf16 MFMA operands, fp32 accumulate/statistics (BASELINE.json configs[4]); "
                         "the headline metric is fp32")
    ap.add_argument("--job-samples", type=int, default=2000,
                    help="samples of the `main.py generate` job timed under end_to_end (batch 32, files on disk; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short measurements of configs[3]/[4] and the end-to-end rate")
    ap.add_argument("--allow-blocking", action="store_true",
                    help="N>1: fall back to the blocking gather if the overlapped one is refused (default: exit non-zero)")
    ap.add_argument("--layers", action="store_true", help="per-layer kernel breakdown on stderr")
    ap.add_argument("--stub-producer", action="store_true",
                    help="CPU rehearsal of the N-rank control flow: gloo, deterministic uint8 pairs instead of the kernels "
                         "(the line says data: stub and is not a measurement)")
    args = ap.parse_args()
    t_proc = time.perf_counter()

    if args.gpus > 1 and not os.environ.get("WORLD_SIZE"):
        # no launcher: start the N ranks ourselves (before anything in this process imports torch or touches HIP)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from gan_segmentation_amd import dist as gdist

    stub = args.stub_producer
    rank, world, local_rank = gdist.init_from_env("gloo" if stub else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the two must agree (torch.distributed.run --nproc-per-node %d, or no "
                         "launcher at all: bench.py then starts its own ranks)" % (args.gpus, world, args.gpus))
    if stub:
        dev = torch.device("cpu")
        my_device = "cpu (stub producer, gloo)"
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (the generate path has no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("rank %d: LOCAL_RANK %d but only %d HIP devices are visible" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        prop = torch.cuda.get_device_properties(local_rank)
        my_device = "cuda:%d %s (%s, %d CUs)" % (local_rank, prop.name, getattr(prop, "gcnArchName", "?"), prop.multi_processor_count)
    # what the process group really is -- not what --gpus asked for: rank count and every rank's device, gathered ONCE
    # before the timed region
    ranks = dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1
    devices = [my_device]
    if ranks > 1:
        devices = [None] * ranks
        dist.all_gather_object(devices, my_device)
    launcher = ("self-launched by bench.py" if os.environ.get("GSA_BENCH_SELF_LAUNCHED") else
                "external (torch.distributed.run)" if world > 1 else "none")

    B = args.batch
    if stub:
        R, nch = 32, 3
        gat = gdist.PairGatherer(B, R, nch, device=dev, dst=0, depth=2)
        producer = StubProducer(rank, B, R, nch)
        loop = TimedLoop(producer, gat, world, dev, args.allow_blocking)
        timing = loop.run(args.settle_steps + args.warmup, args.steps, args.repeats)
        dt = median([loop.max_over_ranks(d) for d in timing["dts_local"]])
        if rank == 0:
            # content and order of the last gathered batch: rank r's row must hold rank r's pairs of the last step
            ok = True
            if timing["gather"] != "blocking":
                parts = gat.result(timing["last_slot"])
                for r in range(ranks):
                    wi, wm = StubProducer(r, B, R, nch).pairs(args.settle_steps + args.warmup + args.steps * max(1, args.repeats) - 1)
                    ok = ok and bool(torch.equal(parts[r][0], wi)) and bool(torch.equal(parts[r][1], wm))
            print(json.dumps({
                "metric": "bench.py control-flow rehearsal (stub producer, NOT a measurement)",
                "value": round(world * B * args.steps / dt, 3), "unit": "stub pairs/s", "n_gpus": args.gpus, "ranks": ranks,
                "devices": devices, "launcher": launcher, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u8", "data": "stub",
                "config": {"workload": "stub producer %dx%dx%d u8 pairs, batch=%d per rank" % (R, R, nch, B),
                           "global_batch": world * B, "parallelism": "dp%d" % world, "gather": timing["gather"],
                           "backend": dist.get_backend() if ranks > 1 else "none"},
                "gathered_matches_producers": ok}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.image_generator import ImageGenerator
    os.environ.setdefault("GSA_GRAPH", "0")     # the headline loop is eager by construction (fp32 batch 8 never replays a graph); pinned and printed
    mr = W.GAN_MAX_RES_LOG2[args.gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[local_rank], batch_size=args.batch,
                                     precision=args.precision)
    # per-rank inputs keyed by global sample index, resident in HBM before the timed region
    z, noise = W.synthetic_inputs(gcfg, B, seed_z=1000 + rank, seed_noise=2000 + rank)
    z = torch.from_numpy(z).to(dev)
    noise = [torch.from_numpy(a).to(dev) for a in noise]
    ctx = gen.netG._model.ctx

    # the pairs of every rank land on rank 0 through ONE gather per batch, double buffered so that the transfer
    # of batch k overlaps the kernels of batch k+1 (dist.PairGatherer); at N=1 the outputs simply stay in HBM
    gat = gdist.PairGatherer(B, 2 ** mr, gcfg["channels"], device=dev, dst=0, depth=2)

    def produce(out):
        return gen.generate_batch(z, noise, out=out) if out is not None else gen.generate_batch(z, noise)

    loop = TimedLoop(produce, gat, world, dev, args.allow_blocking, sync=torch.cuda.synchronize)
    t_setup = time.perf_counter() - t_proc
    # ---- timed region: EXACTLY K steps (no per-launch events here: 2 event packets around each of the
    # ~130 launches of a step cost ~5 % of the step)
    timing = loop.run(args.settle_steps + args.warmup, args.steps, args.repeats)
    t_warm, state, last_slot = timing["warmup_s"], {"gather": timing["gather"]}, timing["last_slot"]
    # ---- roofline pass (untimed): the same K steps with every launch bracketed by HIP events on its
    # stream, and with the decoder/synthesis stream overlap off, so that a kernel's duration is its own
    # and not stretched by the kernel running beside it
    t1 = time.perf_counter()
    ctx.set_overlap(0)
    ctx.profile_enable(2 if args.layers else 1)
    ctx.profile_reset()
    for _ in range(args.steps):
        gen.generate_batch(z, noise)
    torch.cuda.synchronize()
    entries = ctx.profile_entries()
    ctx.profile_enable(0)
    ctx.set_overlap(-1)        # back to the default (by batch size)
    t_prof = time.perf_counter() - t1
    dts = [loop.max_over_ranks(d) for d in timing["dts_local"]]      # per region: the slowest rank
    dt = median(dts)

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / dt
        entries.sort(key=lambda e: -e["ms"])
        top = entries[0]
        kms = sum(e["ms"] for e in entries)
        # FLOP the kernels really execute (Winograd forms, the sub-pixel form of nearest-x2 + conv3x3, 1x1 shortcuts at input
        # resolution): what the MFMA roofline of the whole step is priced on
        exec_gflop = sum(e["flops"] for e in entries) / args.steps / B / 1e9
        roofline = roofline_of(top, args.precision, pmc_traffic(top["name"]))
        mfma_peak = PEAK_FP32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS
        t_mfma, t_hbm = roof_seconds(entries, args.precision)
        roof_ms = 1e3 * (t_mfma + t_hbm) / args.steps
        step_ms = 1e3 * dt / args.steps
        out_img, out_mask = gat.buffers(last_slot) if state["gather"] != "blocking" else gen.generate_batch(z, noise)
        out = {
            "metric": "synthetic (image,mask) pairs/sec, %s-%d StyleGAN+decoder" % (args.gan.upper(), 2 ** mr),
            "value": round(value, 3), "unit": "pairs/s", "n_gpus": world, "ranks": ranks, "devices": devices, "launcher": launcher,
            "steps": args.steps,
            "warmup": args.warmup, "settle_steps": args.settle_steps, "ms_per_step": round(step_ms, 3), "higher_is_better": True,
            # the spread of the headline: `repeats` regions of exactly `steps` steps each, back to back after one warm-up; value and
            # ms_per_step are the MEDIAN region's, runs[] holds every region in the order it was timed
            "repeats": len(dts), "runs": [round(world * B * args.steps / d, 3) for d in dts],
            "min": round(world * B * args.steps / max(dts), 3), "max": round(world * B * args.steps / min(dts), 3),
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "stylegan-%s %d^2 synthesis + %d-class decoder, batch=%d per GPU, %s; "
                                   "synthetic weights/latents/noise; (img u8, mask u8) resident on rank 0"
                                   % (args.gan, 2 ** mr, dcfg["num_classes"], B,
                                      "fp32 (BASELINE.json configs[1])" if args.precision == "fp32" else
                                      "bf16 MFMA operands, fp32 accumulate and statistics (BASELINE.json configs[4])"),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       # which exchange the timed steps contained: none (1 GPU), the overlapped asynchronous gather
                       # (dist.PairGatherer) or the blocking fallback (--allow-blocking only)
                       "gather": state["gather"],
                       "backend": dist.get_backend() if ranks > 1 else "none",
                       "graph": "eager (GSA_GRAPH=%s)" % os.environ.get("GSA_GRAPH", "")},
            "roofline": roofline,
            "whole_path": {
                # the FLOP the kernels EXECUTE per second against the f32 MFMA peak (dense bf16 peak in bf16 mode): the utilisation
                # of the pipe over the whole step; and the step against its own strength-reduced roof
                "executed_gflop_per_sample": round(exec_gflop, 2),
                "executed_tflops": round(exec_gflop * value / 1e3, 2),
                "mfma_peak_tflops": mfma_peak,
                "frac_of_mfma_peak": round(exec_gflop * value / 1e3 / world / mfma_peak, 4),
                "roof_ms": round(roof_ms, 3),
                "roof_ms_mfma": round(1e3 * t_mfma / args.steps, 3), "roof_ms_hbm": round(1e3 * t_hbm / args.steps, 3),
                "frac_of_roof": round(roof_ms / step_ms, 4),
                "roof_note": "roof_ms = per kernel max(executed FLOP / MFMA peak, algorithmic bytes / 6.29 TB/s copy rate), summed over the step",
                # the same step in the reference's formulation (2*MACs of the direct convolutions) -- a throughput, not a utilisation
                "reference_gflop_per_sample": GFLOP_PER_SAMPLE[args.gan],
                "algorithmic_tflops": round(GFLOP_PER_SAMPLE[args.gan] * value / 1e3, 2),
                "algorithmic_frac": round(GFLOP_PER_SAMPLE[args.gan] * value / 1e3 / world / mfma_peak, 4),
                "algorithmic_hbm_gbs": round(MB_PER_SAMPLE[args.gan] * value / 1e3, 1),
                "hbm_frac_of_peak": round(MB_PER_SAMPLE[args.gan] * value / 1e3 / world / PEAK_HBM_GBS, 4),
                "hbm_frac_of_measured_peak": round(MB_PER_SAMPLE[args.gan] * value / 1e3 / world / MEASURED_HBM_GBS, 4),
                "kernel_ms_per_step": round(kms / args.steps, 3),
            },
            # the bytes the timed configuration produced (rank 0, sample 0 of the last timed step) against the C oracle
            "output": output_check(args.gan, B, args.precision, out_img, out_mask),
            "kernels": [{"name": e["name"], "ms_per_step": round(e["ms"] / args.steps, 3),
                         "executed_tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 2) if e["ms"] > 0 else 0.0,
                         "algorithmic_tflops": round(e["alg_flops"] / (e["ms"] * 1e-3) / 1e12, 2) if e["ms"] > 0 else 0.0,
                         "gbs": round(e["bytes"] / (e["ms"] * 1e-3) / 1e9, 1) if e["ms"] > 0 else 0.0}
                        for e in entries[:8]],
        }
        if args.layers:
            for e in entries:
                print("%-70s %8.3f ms/step %7.2f TF/s executed (%7.2f algorithmic) %8.1f GB/s" % (
                    e["name"], e["ms"] / args.steps, e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] else 0,
                    e["alg_flops"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] else 0,
                    e["bytes"] / (e["ms"] * 1e-3) / 1e9 if e["ms"] else 0), file=sys.stderr)
        wall = {"setup": round(t_setup, 2), "warmup": round(t_warm, 3), "timed": round(sum(dts), 4), "timed_median_region": round(dt, 4),
                "profile_pass": round(t_prof, 3)}
        if world == 1 and not args.no_secondary:
            t1 = time.perf_counter()
            out["end_to_end"] = measure_end_to_end(gen, B, max(4, min(args.steps, 20)), dev)
            if args.gan == "ffhq" and args.precision == "fp32" and args.job_samples > 0:
                try:
                    out["end_to_end"].update(measure_generate_job("ffhq", args.job_samples, 32, dev))
                except Exception as e:      # a full or read-only temporary directory must not cost the driver its line
                    out["end_to_end"]["generate_job_error"] = "%s: %s" % (type(e).__name__, e)
            sec = []
            if (args.gan, args.precision, B) != ("bedrooms", "fp32", 64):
                sec.append(measure_secondary("bedrooms", 64, "fp32", 10, 2, dev))      # BASELINE.json configs[3]
            if (args.gan, args.precision, B) != ("cars", "bf16", 4):
                sec.append(measure_secondary("cars", 4, "bf16", 40, 6, dev, graph=True))   # per-GPU share of configs[4], the replayed step
            out["secondary"] = sec
            try:
                out["box"] = measure_box(dev)
                if args.precision == "fp32" and out["roofline"].get("bound") == "mfma" and out["box"].get("sgemm_4096_tflops"):
                    # the dominant kernel's executed rate against what the vendor's own fp32 GEMM reaches on this box (context for `frac`,
                    # which stays priced against the nominal peak)
                    out["box"]["dominant_kernel_over_sgemm"] = round(out["roofline"]["executed_tflops"] / out["box"]["sgemm_4096_tflops"], 4)
            except Exception as e:      # calibration only: never costs the line
                out["box"] = {"error": "%s: %s" % (type(e).__name__, e)}
            wall["secondary"] = round(time.perf_counter() - t1, 2)
        if world == 1 and not args.no_cpu_baseline:
            t1 = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(args.gan)
            wall["cpu_baseline"] = round(time.perf_counter() - t1, 2)
        out["wall_s"] = wall
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
