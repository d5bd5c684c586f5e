"""Tile rate over a whole level of the reference's image pyramid (level n = n x n DataChunk tiles of
[-2,2]^2): kernel time, pinned D2H time and wall tiles/s for one GPU context.
    python scripts/level_rate.py [level] [mrd]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from distributedmandelbrot_amd import MandelbrotDevice

level = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mrd = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = MandelbrotDevice(0)
for item in sys.argv[3:]:                      # library options, e.g. heavy_share=0 (always "group") / 65536 (always "scan")
    k, _, v = item.partition("=")
    dev.set_option(k, int(v))
pin = dev.pinned_empty((16777216,), np.uint8)
dev.datachunk(level, mrd, 0, 0, out_bytes=pin)  # warm-up
ks, ds, its, never, imm, rle = [], [], 0, 0, 0, 0
t0 = time.perf_counter()
for ir in range(level):
    for ii in range(level):
        _, _, st = dev.datachunk(level, mrd, ir, ii, out_bytes=pin)
        ks.append(st.kernel_ms); ds.append(st.d2h_ms); its += st.pixel_iterations
        never += st.all_bytes_zero; imm += st.all_bytes_one; rle += (1 + 5 * st.rle_runs < 1 + 16777216)
dt = time.perf_counter() - t0
n = level * level
print(f"level {level} mrd {mrd}: {n} tiles in {dt:.3f} s = {n/dt:.1f} tiles/s ({its/dt/1e9:.0f} G pixel-iter/s wall); "
      f"kernel ms mean {np.mean(ks):.3f} median {np.median(ks):.3f} max {np.max(ks):.3f} (sum {np.sum(ks)/1e3:.3f} s); "
      f"d2h ms mean {np.mean(ds):.3f}; Never {never} Immediate {imm} RLE-smaller {rle} of {n}")

# the same level with two tiles in flight (mbk_datachunk_submit / mbk_wait): D2H of tile n overlaps kernel n+1
pins = [dev.pinned_empty((16777216,), np.uint8) for _ in range(2)]
tiles = [(ir, ii) for ir in range(level) for ii in range(level)]
t0 = time.perf_counter()
dev.submit_datachunk(0, level, mrd, *tiles[0], pins[0])
for i in range(1, len(tiles) + 1):
    if i < len(tiles):
        dev.submit_datachunk(i % 2, level, mrd, *tiles[i], pins[i % 2])
    dev.wait((i - 1) % 2)
dt2 = time.perf_counter() - t0
print(f"two slots in flight: {n} tiles in {dt2:.3f} s = {n/dt2:.1f} tiles/s")

# ... and without copying uniform tiles off the GPU (MBK_LAZY_UNIFORM): what the pipelined worker does
t0 = time.perf_counter()
skipped = 0
dev.submit_datachunk(0, level, mrd, *tiles[0], pins[0], lazy_uniform=True)
for i in range(1, len(tiles) + 1):
    if i < len(tiles):
        dev.submit_datachunk(i % 2, level, mrd, *tiles[i], pins[i % 2], lazy_uniform=True)
    st = dev.wait((i - 1) % 2)
    skipped += st.all_bytes_zero or st.all_bytes_one
dt3 = time.perf_counter() - t0
print(f"two slots in flight, uniform tiles not copied ({skipped} of {n}): {n} tiles in {dt3:.3f} s = {n/dt3:.1f} tiles/s")
