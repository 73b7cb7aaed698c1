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
    """Threads this process may really use: CPU affinity capped by the cgroup CPU quota; on a
    shared GPU box that reports neither, the documented one-GPU CPU share (16)."""
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
    return n if n <= 32 else 16


def cpu_baseline(gan, seconds_budget=25.0):
    """Stand-in for the reference's mxnet-CPU path (MXNet is not installable here): the
    semantic torch-CPU restatement (oneDNN convolutions), batch 1, all host cores."""
    import torch
    from gan_segmentation_amd import weights as W
    from oracle import ref_semantic as S
    mr = W.GAN_MAX_RES_LOG2[gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
    cores = host_cores()
    torch.set_num_threads(cores)
    times = []
    t_start = time.perf_counter()
    i = 0
    while True:
        z, noise = W.synthetic_inputs(gcfg, 1, seed_z=100 + i, seed_noise=200 + i)
        t0 = time.perf_counter()
        S.generate(gcfg, gp, dcfg, dp, z, noise)
        dt = time.perf_counter() - t0
        if i > 0:       # first sample is the warm-up
            times.append(dt)
        i += 1
        if (len(times) >= 1 and time.perf_counter() - t_start > seconds_budget) or len(times) >= 16:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s %d^2 synthesis+decoder, batch 1, %d samples after 1 warm-up, median; torch-CPU fp32 "
                      "(oneDNN) restatement oracle/ref_semantic.py standing in for the reference's mxnet-CPU path"
                      % (gan, 2 ** mr, len(times))}


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


def measure_secondary(gan, batch, precision, steps, warmup, dev):
    """A short, separate measurement of another BASELINE.json configuration (own model, own context), reported under
    `secondary` -- the headline fields stay those of configs[1]."""
    import torch
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.image_generator import ImageGenerator
    mr = W.GAN_MAX_RES_LOG2[gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[dev.index], batch_size=batch, precision=precision)
    z, noise = W.synthetic_inputs(gcfg, batch, seed_z=1000, seed_noise=2000)
    z = torch.from_numpy(z).to(dev)
    noise = [torch.from_numpy(a).to(dev) for a in noise]
    for _ in range(warmup):
        img, mask = gen.generate_batch(z, noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        img, mask = gen.generate_batch(z, noise)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"workload": "stylegan-%s %d^2 synthesis + decoder, batch=%d, %s, 1 GPU" % (gan, 2 ** mr, batch, precision),
           "value": round(batch * steps / dt, 2), "unit": "pairs/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * dt / steps, 3), "dtype": "f32" if precision == "fp32" else "bf16",
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

    def run(self, warmup, steps):
        t1 = time.perf_counter()
        for _ in range(warmup):
            self.step()
        self.fence()
        self.agree_on_fallback()
        t_warm = time.perf_counter() - t1
        t0 = time.perf_counter()
        for _ in range(steps):               # EXACTLY `steps` steps between two fences
            self.step()
        self.fence()
        dt = time.perf_counter() - t0
        return {"warmup_s": t_warm, "dt_local": dt, "gather": self.gather, "last_slot": (self.k - 1) % self.gat.depth}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (default 50: a third of a second of GPU time at FFHQ batch 8)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gan", default="ffhq", choices=("ffhq", "cars", "bedrooms"))
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU per step")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16"),
                    help="bf16 = bf16 MFMA operands, fp32 accumulate/statistics (BASELINE.json configs[4]); "
                         "the headline metric is fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short measurements of configs[3]/[4] and the end-to-end rate")
    ap.add_argument("--allow-blocking", action="store_true",
                    help="N>1: fall back to the blocking gather if the overlapped one is refused (default: exit non-zero)")
    ap.add_argument("--layers", action="store_true", help="per-layer kernel breakdown on stderr")
    args = ap.parse_args()
    t_proc = time.perf_counter()

    import torch
    import torch.distributed as dist
    from gan_segmentation_amd import dist as gdist
    from gan_segmentation_amd import weights as W
    from gan_segmentation_amd.image_generator import ImageGenerator

    rank, world, local_rank = gdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the generate path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    mr = W.GAN_MAX_RES_LOG2[args.gan]
    gcfg, dcfg = W.generator_config(mr), W.decoder_config(mr)
    gp, dp = W.synthetic_generator_params(gcfg, seed=2), W.synthetic_decoder_params(dcfg, seed=3)
    gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[local_rank], batch_size=args.batch,
                                     precision=args.precision)
    B = args.batch
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
    timing = loop.run(args.warmup, args.steps)
    t_warm, dt, state, last_slot = timing["warmup_s"], timing["dt_local"], {"gather": timing["gather"]}, timing["last_slot"]
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
    dt = loop.max_over_ranks(dt)

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / dt
        entries.sort(key=lambda e: -e["ms"])
        top = entries[0]
        kms = sum(e["ms"] for e in entries)
        # FLOP the kernels really execute (the sub-pixel form of nearest-x2 + conv3x3 needs 4 taps instead of 9 and the
        # 1x1 shortcuts run at input resolution): what the MFMA roofline of the whole step is priced on
        exec_gflop = sum(e["flops"] for e in entries) / args.steps / B / 1e9
        # `achieved` follows the bench contract: ALGORITHMIC FLOP per launch (the layer in the reference's formulation: 2*MACs
        # of the direct convolution) / the launch time.  The kernel may execute fewer (Winograd F(2x2,3x3):
        # 16 products per 2x2 outputs instead of 36; sub-pixel form: 4 taps instead of 9): `executed_*` is what the matrix
        # cores really issue, i.e. the utilisation of the MFMA pipe.
        ach = top["alg_flops"] / (top["ms"] * 1e-3) / 1e12 if top["ms"] > 0 else 0.0
        ach_exec = top["flops"] / (top["ms"] * 1e-3) / 1e12 if top["ms"] > 0 else 0.0
        tr = pmc_traffic(top["name"])
        bound, peak, unit = "mfma", PEAK_FP32_TFLOPS, "TFLOP/s"
        if args.precision == "bf16":     # 16x the MFMA rate: the same kernels sit under the HBM roof
            bound, peak, unit = "hbm", PEAK_HBM_GBS, "GB/s"
            ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9 if top["ms"] > 0 else 0.0
        roofline = {"bound": bound, "kernel": top["name"], "achieved": round(ach, 3), "peak": peak,
                    "unit": unit, "frac": round(ach / peak, 4),
                    # what the pipes really do (read these first): the FLOP the matrix cores issue (Winograd / sub-pixel forms execute
                    # fewer than the algorithmic count) against the f32 MFMA peak, and the kernel's algorithmic bytes against HBM
                    "executed_tflops": round(ach_exec, 3), "executed_frac_of_mfma_peak": round(ach_exec / PEAK_FP32_TFLOPS, 4),
                    "hbm_gbs": round(top["bytes"] / (top["ms"] * 1e-3) / 1e9, 1) if top["ms"] > 0 else 0.0,
                    "hbm_frac": round(top["bytes"] / (top["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if top["ms"] > 0 else 0.0,
                    "avg_launch_ms": round(top["ms"] / max(1, top["launches"]), 4),
                    "launches": top["launches"],
                    "note": "frac = ALGORITHMIC FLOP of the layer in the reference's formulation (SURVEY.md 8d) / time / peak, as the bench "
                            "contract prescribes: a throughput in the reference's units that exceeds 1.0 for a Winograd kernel; the "
                            "utilisation of the matrix pipe is executed_frac_of_mfma_peak",
                    "measured": "HIP events around every launch in a second pass of the same K steps, run right "
                                "after the timed region with the decoder-beside-synthesis stream overlap off "
                                "(durations not stretched by a concurrent kernel)",
                    "flops_per_launch": round(top["alg_flops"] / max(1, top["launches"])),
                    "executed_flops_per_launch": round(top["flops"] / max(1, top["launches"])),
                    "algorithmic_bytes_per_launch": round(top["bytes"] / max(1, top["launches"])),
                    "traffic": tr["bytes_per_launch"] if tr else None,
                    "traffic_source": tr["source"] if tr else None}
        mfma_peak = PEAK_FP32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS
        out_img, out_mask = gat.buffers(last_slot) if state["gather"] != "blocking" else gen.generate_batch(z, noise)
        out = {
            "metric": "synthetic (image,mask) pairs/sec, %s-%d StyleGAN+decoder" % (args.gan.upper(), 2 ** mr),
            "value": round(value, 3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "stylegan-%s %d^2 synthesis + %d-class decoder, batch=%d per GPU, %s; "
                                   "synthetic weights/latents/noise; (img u8, mask u8) resident on rank 0"
                                   % (args.gan, 2 ** mr, dcfg["num_classes"], B,
                                      "fp32 (BASELINE.json configs[1])" if args.precision == "fp32" else
                                      "bf16 MFMA operands, fp32 accumulate and statistics (BASELINE.json configs[4])"),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       # which exchange the timed steps contained: none (1 GPU), the overlapped asynchronous gather
                       # (dist.PairGatherer) or the blocking fallback (--allow-blocking only)
                       "gather": state["gather"]},
            "roofline": roofline,
            "whole_path": {
                # reference FLOPs per second against the f32 MFMA peak (dense bf16 peak in bf16 mode) -- and the
                # FLOP the kernels execute after the sub-pixel rewrite, which is the roofline figure of the whole step
                "tflops": round(GFLOP_PER_SAMPLE[args.gan] * value / 1e3, 2),
                "mfma_peak_tflops": mfma_peak,
                "frac_of_mfma_peak": round(GFLOP_PER_SAMPLE[args.gan] * value / 1e3 / world / mfma_peak, 4),
                "reference_gflop_per_sample": GFLOP_PER_SAMPLE[args.gan],
                "executed_gflop_per_sample": round(exec_gflop, 2),
                "executed_tflops": round(exec_gflop * value / 1e3, 2),
                "frac_of_mfma_peak_executed": round(exec_gflop * value / 1e3 / world / mfma_peak, 4),
                "algorithmic_hbm_gbs": round(MB_PER_SAMPLE[args.gan] * value / 1e3, 1),
                "hbm_frac_of_peak": round(MB_PER_SAMPLE[args.gan] * value / 1e3 / world / PEAK_HBM_GBS, 4),
                "hbm_frac_of_measured_peak": round(MB_PER_SAMPLE[args.gan] * value / 1e3 / world / MEASURED_HBM_GBS, 4),
                "kernel_ms_per_step": round(kms / args.steps, 3),
            },
            # the bytes the timed configuration produced (rank 0, sample 0 of the last timed step) against the C oracle
            "output": output_check(args.gan, B, args.precision, out_img, out_mask),
            "kernels": [{"name": e["name"], "ms_per_step": round(e["ms"] / args.steps, 3),
                         "tflops": round(e["alg_flops"] / (e["ms"] * 1e-3) / 1e12, 2) if e["ms"] > 0 else 0.0,
                         "executed_tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 2) if e["ms"] > 0 else 0.0,
                         "gbs": round(e["bytes"] / (e["ms"] * 1e-3) / 1e9, 1) if e["ms"] > 0 else 0.0}
                        for e in entries[:8]],
        }
        if args.layers:
            for e in entries:
                print("%-70s %8.3f ms/step %7.2f TF/s executed (%7.2f algorithmic) %8.1f GB/s" % (
                    e["name"], e["ms"] / args.steps, e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] else 0,
                    e["alg_flops"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] else 0,
                    e["bytes"] / (e["ms"] * 1e-3) / 1e9 if e["ms"] else 0), file=sys.stderr)
        wall = {"setup": round(t_setup, 2), "warmup": round(t_warm, 3), "timed": round(dt, 4), "profile_pass": round(t_prof, 3)}
        if world == 1 and not args.no_secondary:
            t1 = time.perf_counter()
            out["end_to_end"] = measure_end_to_end(gen, B, max(4, min(args.steps, 20)), dev)
            sec = []
            if (args.gan, args.precision, B) != ("bedrooms", "fp32", 64):
                sec.append(measure_secondary("bedrooms", 64, "fp32", 10, 2, dev))      # BASELINE.json configs[3]
            if (args.gan, args.precision, B) != ("cars", "bf16", 4):
                sec.append(measure_secondary("cars", 4, "bf16", 20, 3, dev))           # per-GPU share of configs[4]
            out["secondary"] = sec
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
