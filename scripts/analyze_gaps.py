"""Where does a back-to-back step go beside its tile kernel?  From a `rocprofv3 --kernel-trace` CSV of bench.py: per tile
kernel its duration, the gap to the next tile kernel on the stream, and where the next launch's pre-pass (classify) ran --
when it started and ended relative to the tile kernel it overlaps.
    python scripts/analyze_gaps.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys

import numpy as np

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
name = lambda r: r.get("Kernel_Name") or r.get("Name")
tiles = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r)) for r in rows if "tile_units_kernel" in name(r)]
cls = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "classify_units_kernel" in name(r)]
tiles.sort()
cls.sort()
by_kind = {}
for i in range(len(tiles) - 1):
    s, e, n = tiles[i]
    s2 = tiles[i + 1][0]
    if tiles[i + 1][2] != n or s2 - e > 200000:      # another leg / a pause of the host
        continue
    # the classify that ended between this tile's start and the next tile's start (the next launch's pre-pass)
    c = [(cs, ce) for cs, ce in cls if s < ce <= s2 + 1000]
    rec = by_kind.setdefault(n, [])
    rec.append((e - s, s2 - e, (c[-1][0] - s) if c else np.nan, (c[-1][1] - e) if c else np.nan, (c[-1][1] - c[-1][0]) if c else np.nan))
for n, rec in by_kind.items():
    a = np.array(rec, float) / 1e3
    a = a[len(a) // 4:]                                # skip the ramp
    print(n[:90])
    print(f"   {len(a)} steps: tile kernel {np.mean(a[:, 0]):7.1f} us  gap to the next tile kernel {np.mean(a[:, 1]):6.1f} us (median {np.median(a[:, 1]):5.1f})"
          f"  -> step {np.mean(a[:, 0] + a[:, 1]):7.1f} us")
    print(f"   next launch's classify: started {np.nanmean(a[:, 2]):7.1f} us after this tile kernel began, ended {np.nanmean(a[:, 3]):+6.1f} us relative to its END "
          f"(median {np.nanmedian(a[:, 3]):+6.1f}), start-to-end {np.nanmean(a[:, 4]):6.1f} us")
