"""Analyse gpurun_out/trace_*.bin from profiles/microbench/occupancy_trace.hip."""
import sys
import numpy as np

rec = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("hw", "<u4"), ("xcc", "<u4"), ("blk", "<u4"), ("sum", "<u4")])
for path in sys.argv[1:]:
    r = np.fromfile(path, dtype=rec)
    t0 = r["t0"].astype(np.int64); t1 = r["t1"].astype(np.int64)
    base = t0.min(); t0 -= base; t1 -= base
    T = t1.max()  # 10 ns ticks
    hw = r["hw"]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7  # gfx9 HW_ID layout
    xcc = r["xcc"] & 15
    unit = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_id = unit * 4 + simd
    ids, inv = np.unique(simd_id, return_inverse=True)
    print(f"{path}: {len(r)} waves, duration {T/100:.1f} us, distinct SIMDs seen {len(ids)} distinct CUs {len(np.unique(unit))}")
    dur = (t1 - t0)
    heavy = r["sum"] > 64 * 200
    print(f"  heavy waves (mean count > 200): {heavy.sum()}  their mean duration {dur[heavy].mean()/100:.1f} us; trivial mean {dur[~heavy].mean()/100:.2f} us")
    # per-SIMD busy: union of intervals is expensive; approximate with occupancy histogram over time bins
    nb = 200
    edges = np.linspace(0, T, nb + 1)
    occ = np.zeros((len(ids), nb), np.float32)
    # add each wave's residency to bins (fractional)
    for k in np.nonzero(dur > (T // nb))[0]:  # long waves
        a, b = t0[k], t1[k]
        i0, i1 = int(a * nb // T), min(int(b * nb // T), nb - 1)
        occ[inv[k], i0:i1 + 1] += 1
    heavy_occ = occ
    idle = (heavy_occ < 0.5).mean(axis=0)     # fraction of SIMDs with no long-running wave in that bin
    low = (heavy_occ < 1.5).mean(axis=0)
    print("  time%   SIMDs w/o long wave   SIMDs with <=1 long wave   mean long waves/SIMD")
    for j in range(0, nb, 10):
        print(f"  {100*j/nb:5.0f}   {idle[j:j+10].mean():8.3f}            {low[j:j+10].mean():8.3f}              {heavy_occ[:, j:j+10].mean():6.2f}")
    per_simd_work = np.bincount(inv, weights=r["sum"].astype(np.float64))
    print(f"  pixel-iterations per SIMD: mean {per_simd_work.mean():.3e} min {per_simd_work.min():.3e} max {per_simd_work.max():.3e} (max/mean {per_simd_work.max()/per_simd_work.mean():.3f})")
    last = np.zeros(len(ids)); np.maximum.at(last, inv, t1)
    print(f"  last wave end per SIMD: mean {last.mean()/100:.1f} us  min {last.min()/100:.1f}  max {last.max()/100:.1f}")
