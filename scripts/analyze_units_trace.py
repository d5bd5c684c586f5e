"""Analyse a trace of profiles/microbench/units_trace.hip: where are the idle issue slots of a "units" launch?

    python scripts/analyze_units_trace.py gpurun_out/units_trace_cfg2.bin

Per workgroup: start / end (100 MHz s_memrealtime), SIMD (XCC, SE, SH, CU, SIMD from HW_ID), unit index, class (0 H: probe
alive at 32 steps; 1 M: gone at step 4..31; 2 V: a row unit of light blocks).  Printed: when each class was dispatched, how
many waves and how many H waves a SIMD holds over the launch, when each SIMD / XCD saw its last wave end (the drain), and
the time SIMDs spend with fewer than 2 resident waves (the arbiter needs 2-3 to saturate the fp64 pipe)."""
import sys

import numpy as np

rec = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("hw", "<u4"), ("xcc", "<u4"), ("unit", "<u4"), ("cls", "<u4")])
for path in sys.argv[1:]:
    r = np.fromfile(path, dtype=rec)
    r = r[r["t1"] > 0]
    t0 = r["t0"].astype(np.int64)
    t1 = r["t1"].astype(np.int64)
    base = t0.min()
    t0 = (t0 - base) / 100.0          # us
    t1 = (t1 - base) / 100.0
    T = t1.max()
    hw = r["hw"]
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    xcc = r["xcc"] & 15
    sid = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    ids, inv = np.unique(sid, return_inverse=True)
    cls = r["cls"]
    print(f"{path}: {len(r)} workgroups on {len(ids)} SIMDs, {len(np.unique(xcc))} XCCs; launch {T:.1f} us (first start to last end)")
    for c, name in ((4, "late M"), (0, "H"), (5, "H set."), (1, "M"), (2, "V")):
        m = cls == c
        if m.any():
            d = t1[m] - t0[m]
            print(f"  class {name:6s}: {m.sum():7d} workgroups, started {t0[m].min():7.1f} .. {t0[m].max():7.1f} us, life mean {d.mean():7.2f} median {np.median(d):7.2f} "
                  f"max {d.max():7.1f} us, ended by {t1[m].max():7.1f}")
    # residency per SIMD over time
    nb = 120
    edges = np.linspace(0, T, nb + 1)
    width = T / nb
    occ = np.zeros((len(ids), nb))
    occ_h = np.zeros((len(ids), nb))
    for arr, sel in ((occ, np.ones(len(r), bool)), (occ_h, (cls == 0) | (cls == 5))):
        a = t0[sel]
        b = t1[sel]
        s = inv[sel]
        i0 = np.minimum((a / width).astype(int), nb - 1)
        i1 = np.minimum((b / width).astype(int), nb - 1)
        for k in range(len(a)):          # fractional residency per bin
            if i0[k] == i1[k]:
                arr[s[k], i0[k]] += (b[k] - a[k]) / width
            else:
                arr[s[k], i0[k]] += (edges[i0[k] + 1] - a[k]) / width
                arr[s[k], i1[k]] += (b[k] - edges[i1[k]]) / width
                if i1[k] > i0[k] + 1:
                    arr[s[k], i0[k] + 1:i1[k]] += 1.0
    print("  time us   waves/SIMD  H waves/SIMD   SIMDs with < 2 waves   SIMDs with < 1 wave")
    step = max(1, nb // 24)
    for j in range(0, nb, step):
        sl = slice(j, j + step)
        print(f"  {edges[j]:7.1f}   {occ[:, sl].mean():8.2f}   {occ_h[:, sl].mean():8.2f}        {(occ[:, sl] < 2).mean():8.3f}              {(occ[:, sl] < 1).mean():8.3f}")
    starved = (occ < 2).mean()
    empty = (occ < 1).mean()
    print(f"  SIMD-time with fewer than 2 resident waves: {100 * starved:.1f} % of the launch; with less than one: {100 * empty:.1f} %")
    last = np.zeros(len(ids))
    np.maximum.at(last, inv, t1)
    first = np.full(len(ids), 1e30)
    np.minimum.at(first, inv, t0)
    print(f"  first wave start per SIMD: mean {first.mean():.1f} us, max {first.max():.1f};  last wave end per SIMD: mean {last.mean():.1f}, min {last.min():.1f}, "
          f"max {last.max():.1f} -> mean idle at the end {T - last.mean():.1f} us = {100 * (T - last.mean()) / T:.1f} % of the launch")
    lastH = np.zeros(len(ids))
    np.maximum.at(lastH, inv[cls == 0], t1[cls == 0])
    print(f"  last H wave end per SIMD: mean {lastH.mean():.1f}, min {lastH.min():.1f}, max {lastH.max():.1f}")
    xid = xcc[np.unique(inv, return_index=True)[1]]
    for x in np.unique(xid):
        print(f"    XCC {x}: last wave end mean {last[xid == x].mean():7.1f} max {last[xid == x].max():7.1f};  H workgroups {int(((cls == 0) & (xcc == x)).sum())}, "
              f"sum of H lives {(t1 - t0)[(cls == 0) & (xcc == x)].sum() / 1e3:7.1f} ms")
    # who ends the launch?  the 24 workgroups that ended last, and per class how many ended in the last 5 / 10 / 20 us
    names = {0: "H", 1: "M", 2: "V", 4: "lateM", 5: "Hset"}
    lastdisp = t0.max()
    print(f"  last dispatch at {lastdisp:.1f} us; launch ends {T - lastdisp:.1f} us later.  The 24 workgroups that ended last:")
    for k in np.argsort(t1)[::-1][:24]:
        print(f"    end {t1[k]:7.1f}  start {t0[k]:7.1f}  life {t1[k] - t0[k]:6.1f} us  class {names.get(int(cls[k]), '?'):5s} unit {int(r['unit'][k]):6d}  XCC {int(xcc[k])}")
    for w in (5.0, 10.0, 20.0):
        sel = t1 > T - w
        print(f"  ended in the last {w:4.0f} us: " + ", ".join(f"{names[c]} {int((sel & (cls == c)).sum())}" for c in (4, 0, 5, 1, 2)))
