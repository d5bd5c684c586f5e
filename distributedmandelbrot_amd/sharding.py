"""Sharding one view over the GPUs of a node: row bands + a plain per-GPU work queue.

The escape-time path has no exchange step (every pixel depends only on its own c), so there is no
collective: the shard unit is a contiguous band of rows of the view (contiguous in the output, one
D2H per band).  Cost per band varies by >100x (in-set vs far exterior), hence dynamic assignment from
one shared cursor: `WorkQueue` when the GPUs live in one process (`render_view`), `SharedCursor` (an
int64 in /dev/shm under an fcntl lock) when there is one process per GPU (bench.py --shard bands under
torchrun).  `rank_bands` is the static interleaved split, kept for callers that cannot share a cursor.

The reference shards the same way one level up: the Distributer leases whole 4096x4096 tiles to
whichever worker asks next (Distributer.cs:335-353).
"""
from __future__ import annotations

import fcntl
import mmap
import os
import struct
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


@dataclass(frozen=True)
class Band:
    index: int
    row0: int
    nrows: int


def make_bands(height: int, band_rows: int) -> List[Band]:
    if height <= 0 or band_rows <= 0:
        raise ValueError("height and band_rows must be positive")
    return [Band(i, r, min(band_rows, height - r)) for i, r in enumerate(range(0, height, band_rows))]


def rank_bands(bands: Sequence[Band], rank: int, world_size: int) -> List[Band]:
    """Static interleaved assignment for one-process-per-GPU runs: rank r takes bands r, r+W, ...
    (neighbouring bands cost about the same, so interleaving balances without communication)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return [b for b in bands if b.index % world_size == rank]


class WorkQueue:
    """The 'plain per-GPU work queue': one shared cursor, each GPU feeder pops the next item."""

    def __init__(self, items: Sequence):
        self._items = list(items)
        self._next = 0
        self._lock = threading.Lock()

    def pop(self):
        with self._lock:
            if self._next >= len(self._items):
                return None
            item = self._items[self._next]
            self._next += 1
            return item

    def __len__(self) -> int:
        return len(self._items)


class SharedCursor:
    """One work cursor shared by the processes of a node (one process per GPU under torch.distributed.run):
    an int64 in a /dev/shm file, advanced under an fcntl lock.  This is the cross-process form of
    WorkQueue -- dynamic assignment with no collective and no RCCL: band cost varies >100x, so a static
    split leaves GPUs idle.  `next()` returns 0, 1, 2, ... each exactly once over all
    processes; the caller maps tickets to work items (bench.py: ticket -> (step, band))."""

    def __init__(self, name: str, create: bool):
        root = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        self.path = os.path.join(root, f"mbk_cursor_{name}")
        self._owner = create
        flags = os.O_RDWR | (os.O_CREAT | os.O_TRUNC if create else 0)
        self._fd = os.open(self.path, flags, 0o600)
        if create:
            os.write(self._fd, struct.pack("<q", 0))
        self._map = mmap.mmap(self._fd, 8)

    def next(self, count: int = 1) -> int:
        fcntl.lockf(self._fd, fcntl.LOCK_EX)
        try:
            (v,) = struct.unpack_from("<q", self._map, 0)
            struct.pack_into("<q", self._map, 0, v + count)
            return v
        finally:
            fcntl.lockf(self._fd, fcntl.LOCK_UN)

    def next_guided(self, limit: int, divisor: int, period: int = 0) -> Tuple[int, int]:
        """Guided self-scheduling: take max(1, remaining // divisor) tickets at once, (first, count) -- large chunks
        while plenty is left, single tickets at the end, so that a consumer whose efficiency grows with the size of a
        launch (a row band of a deep zoom is bounded by its slowest 8x8 block) is not fed crumbs all the way.  A chunk
        never crosses a multiple of `period` (bench.py: the bands of one image).  (limit, 0) when nothing is left."""
        fcntl.lockf(self._fd, fcntl.LOCK_EX)
        try:
            (v,) = struct.unpack_from("<q", self._map, 0)
            if v >= limit:
                return limit, 0
            k = max(1, (limit - v) // max(1, divisor))
            if period > 0:
                k = min(k, period - v % period)
            k = min(k, limit - v)
            struct.pack_into("<q", self._map, 0, v + k)
            return v, k
        finally:
            fcntl.lockf(self._fd, fcntl.LOCK_UN)

    def reset(self, value: int = 0) -> None:
        fcntl.lockf(self._fd, fcntl.LOCK_EX)
        try:
            struct.pack_into("<q", self._map, 0, value)
        finally:
            fcntl.lockf(self._fd, fcntl.LOCK_UN)

    def close(self) -> None:
        if self._map is not None:
            self._map.close()
            os.close(self._fd)
            self._map = None
            if self._owner:
                try:
                    os.unlink(self.path)
                except OSError:
                    pass


def render_view(devices: Sequence, view, mrd: int, *, band_rows: int = 128, want_counts: bool = True,
                want_bytes: bool = True, kernel: str = "default", out_counts: Optional[np.ndarray] = None,
                out_bytes: Optional[np.ndarray] = None
                ) -> Tuple[Optional[np.ndarray], Optional[np.ndarray], List[dict]]:
    """Compute a whole view on several GPUs of this process: one host thread per device pulling row
    bands from a shared WorkQueue, two bands in flight per device (submit_view / wait), every band
    DMA'd straight into its rows of the final image -- no per-band temporary, no second copy.
    `out_counts` / `out_bytes` (height x width) may be supplied by the caller, e.g. pinned arrays from
    MandelbrotDevice.pinned_empty (4x the D2H rate of pageable memory; they live as long as that device).
    `devices` are MandelbrotDevice-like objects: submit_view(slot, view, mrd, window=..., out_counts=...,
    out_bytes=..., kernel=...) + wait(slot), or -- simpler stand-ins -- just compute_view(...).
    Returns (counts | None, bytes | None, per-device stats)."""
    bands = make_bands(view.height, band_rows)
    queue = WorkQueue(bands)
    shape = (view.height, view.width)
    counts = byts = None
    if want_counts:
        counts = out_counts if out_counts is not None else np.empty(shape, np.int32)
        assert counts.shape == shape and counts.dtype == np.int32 and counts.flags.c_contiguous
    if want_bytes:
        byts = out_bytes if out_bytes is not None else np.empty(shape, np.uint8)
        assert byts.shape == shape and byts.dtype == np.uint8 and byts.flags.c_contiguous
    per_dev = [{"bands": 0, "pixel_iterations": 0, "kernel_ms": 0.0} for _ in devices]
    errors: List[BaseException] = []

    def rows(arr, band):
        return arr[band.row0:band.row0 + band.nrows] if arr is not None else None

    def account(slot: int, st) -> None:
        per_dev[slot]["bands"] += 1
        per_dev[slot]["pixel_iterations"] += st.pixel_iterations
        per_dev[slot]["kernel_ms"] += st.kernel_ms

    def feeder(slot: int) -> None:
        dev = devices[slot]
        try:
            if not hasattr(dev, "submit_view"):      # synchronous stand-in: still no temporary
                while True:
                    band = queue.pop()
                    if band is None:
                        return
                    _, _, st = dev.compute_view(view, mrd, window=(0, band.row0, view.width, band.nrows),
                                                want_counts=want_counts, want_bytes=want_bytes, kernel=kernel,
                                                out_counts=rows(counts, band), out_bytes=rows(byts, band))
                    account(slot, st)
            busy = [False, False]
            s = 0
            while True:
                band = queue.pop()
                if band is not None:
                    if busy[s]:
                        account(slot, dev.wait(s))
                    dev.submit_view(s, view, mrd, window=(0, band.row0, view.width, band.nrows),
                                    out_counts=rows(counts, band), out_bytes=rows(byts, band), kernel=kernel)
                    busy[s] = True
                    s ^= 1
                else:
                    for k in (s, s ^ 1):
                        if busy[k]:
                            account(slot, dev.wait(k))
                            busy[k] = False
                    return
        except BaseException as e:
            errors.append(e)

    threads = [threading.Thread(target=feeder, args=(i,), daemon=True) for i in range(len(devices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return counts, byts, per_dev
