"""Asynchronous dataset writer for `generate` (the row after the hot path).

The reference writes every pair from its single Python thread with
``cv2.imwrite(img_%06d.jpg, img[:, :, ::-1])`` / ``cv2.imwrite(mask_%06d.png, mask[:, :, 0])``
(reference main.py:100-103) -- roughly 100 pairs/s at 1024 px, far below what one MI355X now
produces.  Here the uint8 results leave the GPU through pinned ring buffers on a copy stream and are
encoded by a pool of worker threads (PIL/libjpeg/zlib release the GIL), while the GPU computes the
next batch.  File names, formats and contents follow the reference: RGB JPEG quality 95 (what the
BGR flip + cv2 produce), single-channel PNG holding the class index, directory
``BASE_DIR/dataset/train_generated`` (consumer contract:
reference deeplabv3plus/lib/data/segmentation/ffhq_hair_segmentation.py:24-49).

``gpu_jpeg=True`` (what `main.py generate` uses): the JPEG is produced on the GPU by ``jpeg.JpegEncoder``
(csrc/gsa_jpeg.hip -- libjpeg's integer arithmetic, so the file decodes to exactly the pixels the reference's
cv2 call would store) right behind the generate kernels, only the compressed bytes (~15 % of the pixels) cross
PCIe, and the worker threads are left with ``write()`` and the mask PNGs: 14 ms of a host core per 1024^2 image
become 0.03 ms of GPU time.
"""
import os
import queue
import struct
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def default_workers():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 32))


def _png_chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def mask_png_bytes(mask):
    """(H,W) uint8 class indices -> an 8-bit greyscale PNG, deflated at level 1 with the run-length strategy -- the
    settings cv2.imwrite uses by default (IMWRITE_PNG_COMPRESSION 1, IMWRITE_PNG_STRATEGY_RLE), which fit masks
    (long constant runs): 2 ms and 7 KB for a 1024^2 blob mask against 8 ms and 14 KB for zlib's default strategy
    behind PIL.  Filter type 0 on every row; lossless either way (the tests decode it back)."""
    mask = np.ascontiguousarray(mask, np.uint8)
    if mask.ndim != 2:
        raise ValueError("mask must be (H, W)")
    H, W = mask.shape
    rows = np.zeros((H, W + 1), np.uint8)            # a filter-type byte in front of every scanline
    rows[:, 1:] = mask
    comp = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    data = comp.compress(rows.tobytes()) + comp.flush()
    return (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0))
            + _png_chunk(b"IDAT", data) + _png_chunk(b"IEND", b""))


def _write_mask(dst_dir, index, mask):
    with open(os.path.join(dst_dir, "mask_%06d.png" % index), "wb") as f:
        f.write(mask_png_bytes(mask))


def write_pair(dst_dir, index, img, mask, jpeg_quality=95):
    """One (image, mask) pair -> img_%06d.jpg + mask_%06d.png (reference main.py:100-103)."""
    from PIL import Image
    Image.fromarray(img, "RGB").save(os.path.join(dst_dir, "img_%06d.jpg" % index), quality=jpeg_quality)
    _write_mask(dst_dir, index, mask)


def write_encoded_pair(dst_dir, index, jpeg_parts, img, png_stream, mask, H, W, jpeg_quality=95):
    """One pair whose image and/or mask were already compressed on the GPU: ``jpeg_parts`` = (header, scan) or None
    (then ``img`` is encoded here), ``png_stream`` = the zlib stream of the mask or None (then ``mask`` is encoded here)."""
    if jpeg_parts is not None:
        with open(os.path.join(dst_dir, "img_%06d.jpg" % index), "wb") as f:
            f.write(jpeg_parts[0])
            f.write(jpeg_parts[1])
    else:
        from PIL import Image
        Image.fromarray(img, "RGB").save(os.path.join(dst_dir, "img_%06d.jpg" % index), quality=jpeg_quality)
    if png_stream is not None:
        from .png import png_file
        with open(os.path.join(dst_dir, "mask_%06d.png" % index), "wb") as f:
            f.write(png_file(H, W, png_stream))
    else:
        _write_mask(dst_dir, index, mask)


class DeviceCheckFailed(RuntimeError):
    """A batch's status snapshot (``ImageGenerator.snapshot_status``) was non-zero: ``first_index`` is the global index of
    the first sample whose files were withheld; every file below it was checked clean."""

    def __init__(self, first_index, words, after=None):
        self.first_index, self.words = int(first_index), tuple(int(w) for w in words)
        if after is not None:
            # a LATER batch of a run that already failed: its own words were never looked at -- it is withheld because of the
            # first failure, whose cause `after` carries (no cause is made up for this batch)
            self.first_failure = after
            super().__init__("batch starting at global sample index %d withheld: the run failed its device-side check at index %d (%s)"
                             % (self.first_index, after.first_index, after))
            return
        self.first_failure = self
        what = []
        if self.words[0]:
            what.append("an instance-norm statistic left the range of its 64-bit fixed-point sum")
        if self.words[1]:
            what.append("the fused mapping network timed out waiting for a partner workgroup")
        super().__init__("device-side check failed in the batch starting at global sample index %d (%s): the files from "
                         "index %d on were not written, everything below is valid" % (self.first_index, "; ".join(what), self.first_index))


STATUS_RING_DEPTH = 32      # pinned status slots ImageGenerator.snapshot_status() cycles through: more batches than this may never be in flight


class DatasetWriter:
    """``submit(img, mask, first_index)`` returns immediately; ``close()`` waits for every file."""

    def __init__(self, dst_dir, workers=None, slots=3, jpeg_quality=95, gpu_jpeg=False, jpeg_restart=None, gpu_png=False):
        if not 1 <= slots < STATUS_RING_DEPTH:
            raise ValueError("slots must be in [1, %d): a batch's status slot (ImageGenerator.snapshot_status) is reused after "
                             "%d later batches" % (STATUS_RING_DEPTH, STATUS_RING_DEPTH))
        self.dst_dir = dst_dir
        os.makedirs(dst_dir, exist_ok=True)
        self.jpeg_quality = jpeg_quality
        self.gpu_jpeg = gpu_jpeg
        self.gpu_png = gpu_png
        self.jpeg_restart = jpeg_restart
        self._encoders = [{} for _ in range(slots)]   # per slot: {"jpeg": JpegEncoder, "png": PngEncoder} (own output buffers)
        self._stage = [{} for _ in range(slots)]      # per slot and kind: pinned staging for the compressed bytes
        self.pool = ThreadPoolExecutor(max_workers=workers or default_workers())
        self.slots = slots
        self._free = queue.Queue()
        for i in range(slots):
            self._free.put(i)
        self._host = [None] * slots          # pinned (img, mask) buffers, allocated on first use
        self._pending = queue.Queue()
        self._errors = []
        self._lock = threading.Lock()
        self._copy_stream = None
        self._status = {}                    # first_index -> pinned status slots of that batch (or None)
        self._halted = None                  # the DeviceCheckFailed of the first batch that failed its device-side check: nothing from there on is written
        self._dispatcher = threading.Thread(target=self._dispatch, daemon=True)
        self._dispatcher.start()
        self.written = 0
        self.submitted = 0

    # -- producer side ----------------------------------------------------------------------
    def submit(self, img, mask, first_index, status=None):
        """img (N,R,R,3) u8, mask (N,R,R) u8: torch CUDA tensors (copied asynchronously) or numpy.
        ``status``: the pinned snapshot slots ``ImageGenerator.snapshot_status()`` returned for THIS batch (enqueued on the
        producing stream before this call); the dispatcher reads them when the batch's copies have completed and withholds
        the batch's files -- and every later one -- if a device-side check had failed by then (``DeviceCheckFailed``)."""
        if self._errors:
            raise self._errors[0]
        self.submitted += int(img.shape[0])
        with self._lock:
            self._status[first_index] = status
        if isinstance(img, np.ndarray):
            self._pending.put((None, None, np.ascontiguousarray(img), np.ascontiguousarray(mask), first_index))
            return
        import torch
        slot = self._free.get()               # back-pressure: at most `slots` batches in flight
        n = img.shape[0]
        if self.gpu_jpeg or self.gpu_png:
            return self._submit_encoded(slot, img, mask, first_index)
        buf = self._host[slot]
        if buf is None or buf[0].shape[0] < n or buf[0].shape[1:] != img.shape[1:]:
            buf = (torch.empty(tuple(img.shape), dtype=torch.uint8).pin_memory(),
                   torch.empty(tuple(mask.shape), dtype=torch.uint8).pin_memory())
            self._host[slot] = buf
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=img.device)
        cur = torch.cuda.current_stream(img.device)
        self._copy_stream.wait_stream(cur)    # results are ready when the producing stream gets here
        with torch.cuda.stream(self._copy_stream):
            buf[0][:n].copy_(img, non_blocking=True)
            buf[1][:n].copy_(mask, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        img.record_stream(self._copy_stream)
        mask.record_stream(self._copy_stream)
        self._pending.put((slot, ev, buf[0][:n].numpy(), buf[1][:n].numpy(), first_index))

    def _encoder(self, slot, kind, n, H, W, device):
        enc = self._encoders[slot].get(kind)
        if enc is None or enc.n < n or (enc.H, enc.W) != (H, W) or enc.device != device:
            if kind == "jpeg":
                from . import jpeg
                kw = {} if self.jpeg_restart is None else {"restart": self.jpeg_restart}
                worst = jpeg._api()["gsa_jpeg_max_scan_bytes"](H, W, kw.get("restart", jpeg.DEFAULT_RESTART))
                enc = jpeg.JpegEncoder(n, H, W, device, quality=self.jpeg_quality, out_stride=worst, **kw)
            else:
                from . import png
                enc = png.PngEncoder(n, H, W, device)
            self._encoders[slot][kind] = enc
        return enc

    def _submit_encoded(self, slot, img, mask, first_index):
        """Compress on the GPU right behind the generate kernels; the dispatcher thread then fetches the lengths and
        exactly the compressed bytes.  Whatever is not compressed on the GPU is copied out raw, as in the host path."""
        import torch
        n, H, W = img.shape[0], img.shape[1], img.shape[2]
        cur = torch.cuda.current_stream(img.device)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=img.device)
        if not img.is_contiguous():
            img = img.contiguous()
        if not mask.is_contiguous():
            mask = mask.contiguous()
        def pinned(kind, like):                      # raw pinned buffer for whatever is not compressed on the GPU
            raw = self._stage[slot].get(kind)
            if raw is None or raw.shape[0] < n or raw.shape[1:] != like.shape[1:]:
                raw = torch.empty(tuple(like.shape), dtype=torch.uint8).pin_memory()
                self._stage[slot][kind] = raw
            return raw[:n]

        payload = {"n": n, "H": H, "W": W}
        # the encoders run on the copy stream, beside the next batch's generate kernels: their entropy-coding kernels
        # are latency-bound (one lane per restart interval / row group, a few hundred waves) and hide behind them
        self._copy_stream.wait_stream(cur)
        with torch.cuda.stream(self._copy_stream):
            if self.gpu_jpeg:
                enc = self._encoder(slot, "jpeg", n, H, W, img.device)
                payload["jpeg"] = (enc,) + tuple(enc.encode(img))
            else:
                raw = pinned("raw_img", img)
                raw.copy_(img, non_blocking=True)
                payload["img"] = raw.numpy()
            if self.gpu_png:
                enc = self._encoder(slot, "png", n, H, W, img.device)
                payload["png"] = (enc,) + tuple(enc.encode(mask))
            else:
                raw = pinned("raw_mask", mask)
                raw.copy_(mask, non_blocking=True)
                payload["mask"] = raw.numpy()
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        img.record_stream(self._copy_stream)
        mask.record_stream(self._copy_stream)
        self._pending.put((slot, ev, payload, None, first_index))

    def _fetch_encoded(self, slot, kind, out, lengths):
        """-> list of numpy views (one per sample) of the compressed bytes in pinned memory."""
        import torch
        n = out.shape[0]
        with torch.cuda.stream(self._copy_stream):
            ln = lengths.to("cpu", non_blocking=False).numpy().astype(np.int64)     # 4 bytes per sample; synchronises
            if (ln <= 0).any():
                raise RuntimeError("%s encoder overflow (lengths %s)" % (kind, ln))  # impossible with the worst-case stride
            offs = np.concatenate([[0], np.cumsum(ln)])
            stage = self._stage[slot].get(kind)
            if stage is None or stage.numel() < offs[-1]:
                stage = torch.empty(int(max(offs[-1] * 3 // 2, 1 << 16)), dtype=torch.uint8).pin_memory()
                self._stage[slot][kind] = stage
            for i in range(n):
                stage[offs[i]:offs[i + 1]].copy_(out[i, :ln[i]], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        ev.synchronize()
        host = stage.numpy()
        return [host[offs[i]:offs[i + 1]] for i in range(n)]

    # -- consumer side ----------------------------------------------------------------------
    def _dispatch(self):
        while True:
            item = self._pending.get()
            if item is None:
                return
            slot, ev, img, mask, first = item
            futs = []
            try:
                if ev is not None:
                    ev.synchronize()
                # the copy stream waited for the producing stream, so the 8-byte status copy enqueued there before submit()
                # has landed: look at it BEFORE any file of this batch exists
                with self._lock:
                    status = self._status.pop(first, None)
                    halted = self._halted
                if halted is not None:
                    raise DeviceCheckFailed(first, halted.words, after=halted)
                if status is not None:
                    words = [int(v) for t in status for v in t.tolist()]
                    if any(words):
                        failure = DeviceCheckFailed(first, (any(words[0::2]), any(words[1::2])))
                        with self._lock:
                            self._halted = failure      # the first failing batch and ITS words: later batches refer to it
                        raise failure
                if isinstance(img, dict):         # (partly) compressed on the GPU
                    p = img
                    scans = self._fetch_encoded(slot, "jpeg", *p["jpeg"][1:]) if "jpeg" in p else None
                    streams = self._fetch_encoded(slot, "png", *p["png"][1:]) if "png" in p else None
                    futs = [self.pool.submit(write_encoded_pair, self.dst_dir, first + i,
                                             (p["jpeg"][0].header, scans[i]) if scans is not None else None,
                                             p["img"][i] if "img" in p else None,
                                             streams[i] if streams is not None else None,
                                             p["mask"][i] if "mask" in p else None, p["H"], p["W"], self.jpeg_quality)
                            for i in range(p["n"])]
                else:
                    futs = [self.pool.submit(write_pair, self.dst_dir, first + i, img[i], mask[i], self.jpeg_quality)
                            for i in range(img.shape[0])]
            except Exception as e:   # surfaced by the next submit()/close()
                self._errors.append(e)
            # The dispatcher does not wait for the files: the batch's slot (its pinned buffers) is released by the
            # last of its futures, so the files of up to `slots` batches are encoded side by side.
            self._release_when_done(slot, futs)

    def _release_when_done(self, slot, futs):
        if not futs:
            if slot is not None:
                self._free.put(slot)
            return
        state = {"left": len(futs)}

        def done(f):
            exc = f.exception()
            with self._lock:
                if exc is not None:
                    self._errors.append(exc)
                else:
                    self.written += 1
                state["left"] -= 1
                last = state["left"] == 0
            if last and slot is not None:
                self._free.put(slot)

        for f in futs:
            f.add_done_callback(done)

    def drain(self, timeout=600.0):
        """Block until every pair submitted so far is on disk (the writer stays open)."""
        import time
        t0 = time.perf_counter()
        while True:
            with self._lock:
                # any recorded error ends the wait at once: one batch-level failure stands for N pairs that will never be
                # counted, so "written + errors >= submitted" alone would spin until the timeout and hide the real exception
                done = bool(self._errors) or self.written >= self.submitted
            if done:
                break
            if time.perf_counter() - t0 > timeout:
                raise TimeoutError("DatasetWriter.drain: %d of %d pairs written" % (self.written, self.submitted))
            time.sleep(0.0005)
        if self._errors:
            raise self._errors[0]
        return self.written

    def close(self):
        self._pending.put(None)
        self._dispatcher.join()
        self.pool.shutdown(wait=True)
        if self._errors:
            raise self._errors[0]
        return self.written

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
