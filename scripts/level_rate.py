"""Tile rate over a whole level of the reference's image pyramid (level n = n x n DataChunk tiles of
[-2,2]^2): kernel time, pinned D2H time and wall tiles/s for one GPU context -- synchronous, and with 2 .. MBK_SLOTS tiles
in flight, with and without MBK_LAZY_UNIFORM (the worker's mode).
    python scripts/level_rate.py [level] [mrd] [option=value ...]"""
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
print(f"  bound of a perfect pipeline: sum of max(kernel, d2h) per tile = {sum(max(a, b) for a, b in zip(ks, ds)):.1f} ms "
      f"-> {n / sum(max(a, b) for a, b in zip(ks, ds)) * 1e3:.0f} tiles/s; sum of d2h alone {sum(ds):.1f} ms")

nslots = dev.SLOTS
pins = [dev.pinned_empty((16777216,), np.uint8) for _ in range(nslots)]
tiles = [(ir, ii) for ir in range(level) for ii in range(level)]
# host time per tile of the two calls, by kind of tile (MBK_LAZY_UNIFORM, all slots in flight) -- after one untimed pipelined pass:
# the first tile on a slot's stream creates its scratch (dispatch lists, an auxiliary stream, events: milliseconds, once), which
# round 5's figure for the copied tiles (446 us per submit) was mostly made of
for i in range(n + nslots):
    if i >= nslots:
        dev.wait((i - nslots) % nslots)
    if i < n:
        dev.submit_datachunk(i % nslots, level, mrd, *tiles[i], pins[i % nslots], lazy_uniform=True)
acc = {}
inflight_kind = {}
for i in range(n + nslots):
    if i >= nslots:
        t0 = time.perf_counter()
        st = dev.wait((i - nslots) % nslots)
        dt = time.perf_counter() - t0
        kind = "host-answered" if st.kernel_ms == 0.0 else "uniform, computed" if (st.all_bytes_zero or st.all_bytes_one) else "copied"
        a = acc.setdefault(kind, [0, 0.0, 0.0])
        a[0] += 1; a[1] += inflight_kind.pop(i - nslots); a[2] += dt
    if i < n:
        t0 = time.perf_counter()
        dev.submit_datachunk(i % nslots, level, mrd, *tiles[i], pins[i % nslots], lazy_uniform=True)
        inflight_kind[i] = time.perf_counter() - t0
for kind, (k, ts, tw) in sorted(acc.items()):
    print(f"  host time per tile, {kind:18s}: {k:4d} tiles, submit {ts / k * 1e6:6.1f} us, wait {tw / k * 1e6:6.1f} us")
for lazy in (False, True):
    for k in range(2, nslots + 1):
        best, uniform, its2 = None, 0, 0
        for rep in range(2):
            t0 = time.perf_counter()
            uniform, its2 = 0, 0
            for i in range(n + k):
                if i >= k:
                    st = dev.wait((i - k) % k)
                    uniform += st.all_bytes_zero or st.all_bytes_one
                    its2 += st.pixel_iterations
                if i < n:
                    dev.submit_datachunk(i % k, level, mrd, *tiles[i], pins[i % k], lazy_uniform=lazy)
            d = time.perf_counter() - t0
            best = d if best is None else min(best, d)
        assert its2 == its, (its2, its)
        print(f"{k} in flight{', MBK_LAZY_UNIFORM (' + str(uniform) + ' uniform tiles not copied)' if lazy else ''}: "
              f"{n} tiles in {best:.4f} s = {n/best:.1f} tiles/s")

# two contexts on the one GPU, a host thread each (what `worker ADDR PORT 0,0` runs: two feeders): the host's per-tile work
# -- a dozen HIP calls -- is what bounds the all-exterior stretches of a level, and it parallelises
import threading
dev2 = MandelbrotDevice(0)
for item in sys.argv[3:]:
    k, _, v = item.partition("=")
    dev2.set_option(k, int(v))
pins2 = [dev2.pinned_empty((16777216,), np.uint8) for _ in range(nslots)]


def half(d, pp, mine, out):
    k, its3 = nslots, 0
    m = len(mine)
    for i in range(m + k):
        if i >= k:
            its3 += d.wait((i - k) % k).pixel_iterations
        if i < m:
            d.submit_datachunk(i % k, level, mrd, *mine[i], pp[i % k], lazy_uniform=True)
    out.append(its3)


for rep in range(2):
    out = []
    th = [threading.Thread(target=half, args=(dev, pins, tiles[0::2], out)), threading.Thread(target=half, args=(dev2, pins2, tiles[1::2], out))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    d = time.perf_counter() - t0
    assert sum(out) == its
print(f"two contexts x {nslots} in flight, two host threads, MBK_LAZY_UNIFORM: {n} tiles in {d:.4f} s = {n/d:.1f} tiles/s")
