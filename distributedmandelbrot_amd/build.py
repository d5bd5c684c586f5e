"""Build libmbk_hip.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m distributedmandelbrot_amd.build [--force] [--save-temps]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
-ffp-contract=off is MANDATORY: hipcc contracts a*b+c into v_fma_f64 by default, which would change
iteration counts (SURVEY.md probe P2).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libmbk_hip.so")
SOURCES = [os.path.join(CSRC, "mbk_api.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "mbk_kernels.h"), os.path.join(CSRC, "mbk_refill.h"), os.path.join(CSRC, "mbk_loops.inc"), os.path.join(CSRC, "mbk_persist.h"), os.path.join(CSRC, "mbk_scan.h"), os.path.join(CSRC, "mbk_units.h"), os.path.join(CSRC, "mbk_spill.h"), os.path.join(CSRC, "mbk_feeder.h"),
                  os.path.join(os.path.dirname(HERE), "include", "mbk.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fPIC", "-shared", "-Wall", "-Wno-unused-result"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libmbk_hip.so")
    return exe


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force: bool = False, save_temps: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    cmd = [hipcc()] + FLAGS + SOURCES + ["-o", SO]
    cwd = HERE
    if save_temps:   # intermediate .s / .bc files go to the git-ignored build/ directory
        cwd = os.path.join(HERE, "build")
        os.makedirs(cwd, exist_ok=True)
        cmd += ["-save-temps=cwd", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=cwd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True))
