"""MI355X-native Mandelbrot tile worker: a drop-in for the compute path of
ofsouzap/DistributedMandelbrot's worker (DistributedMandelbrotWorkerCUDA.py).

Layout
------
csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/mbk.h) -> libmbk_hip.so
build.py         hipcc driver (in-tree build)
_lib.py          ctypes binding of the C ABI (fails loudly when the library or a GPU is missing)
device.py        MandelbrotDevice: one GPU context; views, DataChunk tiles, async launches
worker.py        the reference worker's interface (process_workload / do_workload_single / main)
                 speaking the unchanged Distributer TCP protocol, plus a per-GPU work-queue farm
sharding.py      row-band work items and the per-GPU queue used to shard one view over N GPUs

There is no CPU fallback anywhere in this package.
"""
from .device import MandelbrotDevice, MbkError, TileStats, View, device_count  # noqa: F401

__all__ = ["MandelbrotDevice", "MbkError", "TileStats", "View", "device_count"]
