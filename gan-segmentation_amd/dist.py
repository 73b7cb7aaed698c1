"""Batch sharding across the GPUs of one node: one process per GPU, ``torch.distributed``
over RCCL (backend "nccl" on ROCm) or gloo (CPU tests).

The reference shards a batch over its ``ctx`` list with ``split_and_load(even_split=False)``
and gathers through host memory (reference image_generator.py:95-114).  Samples are
independent given (z_i, noise_i), so here rank r simply owns a contiguous slice of the global
sample indices and the only exchange is ONE gather of the uint8 results to rank 0 per batch
(4 MiB per FFHQ sample: 3 MiB image + 1 MiB mask, fused into a single buffer per rank).
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous slice [lo, hi) of ``total`` samples owned by ``rank`` -- the slicing of
    ``split_and_load(even_split=False)``: the first ``total % world`` ranks get one extra."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_ranks():
    """(rank, world, local_rank) from the torchrun environment WITHOUT creating a process group: all a path
    needs whose ranks never talk to each other (``main.py generate`` writing files: ``shard_bounds`` only)."""
    return (int(os.environ.get("RANK") or 0), int(os.environ.get("WORLD_SIZE") or 1),
            int(os.environ.get("LOCAL_RANK") or 0))


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    rank, world, local_rank = env_ranks()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def pack_pairs(img, mask):
    """(n,R,R,3) u8 + (n,R,R) u8 -> one (n, R*R*4) u8 buffer: a single message per rank."""
    n = img.shape[0]
    return torch.cat([img.reshape(n, -1), mask.reshape(n, -1)], dim=1).contiguous()


def unpack_pairs(buf, R, channels=3):
    n = buf.shape[0]
    img = buf[:, :R * R * channels].reshape(n, R, R, channels)
    mask = buf[:, R * R * channels:].reshape(n, R, R)
    return img, mask


def gather_pairs(img, mask, counts=None, dst=0, group=None):
    """Gather every rank's (img, mask) to ``dst`` with one collective.  ``counts[r]`` = samples
    of rank r (defaults to equal shards).  Returns (img, mask) in global sample order on ``dst``
    and (None, None) elsewhere."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return img, mask
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    R, ch = img.shape[1], img.shape[3]
    n = img.shape[0]
    if counts is None:
        counts = [n] * world
    nmax = max(counts)
    buf = pack_pairs(img, mask)
    if n < nmax:   # ragged last batch: pad to the common message size
        pad = torch.zeros((nmax - n, buf.shape[1]), dtype=buf.dtype, device=buf.device)
        buf = torch.cat([buf, pad], dim=0)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None, None
    parts = [unpack_pairs(o[:c], R, ch) for o, c in zip(out, counts)]
    return torch.cat([p[0] for p in parts], dim=0), torch.cat([p[1] for p in parts], dim=0)


class PairGatherer:
    """Overlapped, copy-free form of ``gather_pairs`` for a steady stream of equal batches.

    Each rank owns ``depth`` fused send buffers ``[n*R*R*ch image bytes | n*R*R mask bytes]``; the generate
    kernels write straight into the two views ``buffers(slot)`` returns, so nothing is packed.  ``submit(slot)``
    starts ONE asynchronous gather of that buffer (RCCL runs it on its own stream once the producing kernels
    are done) into row ``rank`` of a preallocated ``(world, bytes)`` tensor per slot on ``dst``; the caller
    goes on to enqueue the next batch and only ``wait(slot)``s before that slot is produced into again, so the
    transfer over xGMI hides behind the next batch's kernels.  ``result(slot)`` on ``dst`` = per-rank views
    ``[(img (n,R,R,ch), mask (n,R,R))]`` in rank order (global sample order), no copy."""

    def __init__(self, n, R, channels=3, device="cpu", dst=0, depth=2, group=None, force_collective=False):
        self.n, self.R, self.ch, self.dst, self.depth, self.group = n, R, channels, dst, depth, group
        # force_collective: issue the gather even in a one-rank group (exercises the RCCL call path on a single GPU)
        self.active = dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.img_bytes = n * R * R * channels
        self.bytes = self.img_bytes + n * R * R
        self.send = [torch.empty(self.bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.recv = [torch.empty((self.world, self.bytes), dtype=torch.uint8, device=device)
                     if (self.active and self.rank == dst) else None for _ in range(depth)]
        self.work = [None] * depth

    def _views(self, flat):
        return (flat[:self.img_bytes].view(self.n, self.R, self.R, self.ch),
                flat[self.img_bytes:].view(self.n, self.R, self.R))

    def buffers(self, slot):
        """(img, mask) views of send buffer ``slot`` -- hand them to ``generate_batch(out=...)``."""
        return self._views(self.send[slot])

    def submit(self, slot):
        if not self.active:
            return
        out = [self.recv[slot][r] for r in range(self.world)] if self.rank == self.dst else None
        self.work[slot] = dist.gather(self.send[slot], out, dst=self.dst, group=self.group, async_op=True)

    def wait(self, slot):
        """The gather last submitted from ``slot`` has completed (stream-ordered on CUDA/HIP tensors)."""
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None

    def wait_all(self):
        for slot in range(self.depth):
            self.wait(slot)

    def result(self, slot):
        if not self.active:
            return [self._views(self.send[slot])]
        if self.rank != self.dst:
            return None
        return [self._views(self.recv[slot][r]) for r in range(self.world)]
