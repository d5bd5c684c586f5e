"""Board power / shader clock trace around a command (is the chip power-limited?).

    python scripts/power_trace.py OUT.json -- python bench.py --workload cfg3 --kernel refill ...

Samples GPU 0 at ~100 Hz through librocm_smi64 (ctypes; no subprocess per sample): socket power, the
energy accumulator, the current sclk level and the power cap.  Falls back to hwmon sysfs files, then to
`rocm-smi --json` (slow), whichever works on the box.  Writes a JSON summary: cap, idle power, power and
clock during the busy part of the run (samples above idle + 20 % of the busy range), the energy between
the first and last busy sample, and -- if the command printed a bench.py line -- joules per
G pixel-iteration.  The raw trace is kept (decimated to <= 2000 points)."""
from __future__ import annotations

import ctypes as C
import glob
import json
import os
import subprocess
import sys
import threading
import time


class Freqs(C.Structure):
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32),
                ("frequency", C.c_uint64 * 33)]


class RsmiSampler:
    name = "librocm_smi64"

    def __init__(self):
        self.lib = C.CDLL("librocm_smi64.so")
        if self.lib.rsmi_init(C.c_uint64(0)) != 0:
            raise RuntimeError("rsmi_init failed")
        self.cap_w = None
        cap = C.c_uint64(0)
        if self.lib.rsmi_dev_power_cap_get(0, 0, C.byref(cap)) == 0:
            self.cap_w = cap.value / 1e6
        self.sample()

    def sample(self):
        p, typ = C.c_uint64(0), C.c_int(0)
        power = None
        if self.lib.rsmi_dev_power_get(0, C.byref(p), C.byref(typ)) == 0:
            power = p.value / 1e6
        f = Freqs()
        clk = None
        if self.lib.rsmi_dev_gpu_clk_freq_get(0, 0, C.byref(f)) == 0 and f.current < 33:
            clk = f.frequency[f.current] / 1e6
        e, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
        energy = None
        if self.lib.rsmi_dev_energy_count_get(0, C.byref(e), C.byref(res), C.byref(ts)) == 0:
            energy = e.value * res.value / 1e6   # counter * resolution (uJ) -> J
        if power is None and clk is None:
            raise RuntimeError("rocm_smi returns neither power nor clock")
        return power, clk, energy


class HwmonSampler:
    name = "hwmon sysfs"

    def __init__(self):
        cands = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
        self.dir = next((d for d in sorted(cands) if os.path.exists(os.path.join(d, "power1_average"))
                         or os.path.exists(os.path.join(d, "power1_input"))), None)
        if not self.dir:
            raise RuntimeError("no amdgpu hwmon with power files")
        self.pfile = next(f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(self.dir, f)))
        self.cap_w = self._read("power1_cap", 1e6)
        self.sample()

    def _read(self, f, div):
        try:
            return int(open(os.path.join(self.dir, f)).read()) / div
        except Exception:
            return None

    def sample(self):
        return self._read(self.pfile, 1e6), self._read("freq1_input", 1e6), None


class CliSampler:
    name = "rocm-smi --json"
    cap_w = None

    def __init__(self):
        self.sample()

    def sample(self):
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(out.stdout).values()))
        power = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
        clk = next((float(v.strip("()Mhz")) for k, v in card.items() if k.startswith("sclk clock speed")), None)
        return power, clk, None


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    sampler = None
    errors = []
    for cls in (RsmiSampler, HwmonSampler, CliSampler):
        try:
            sampler = cls()
            break
        except Exception as e:
            errors.append(f"{cls.__name__}: {e!r}")
    trace, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                trace.append((time.perf_counter(),) + tuple(sampler.sample()))
            except Exception:
                pass
            time.sleep(0.008)

    th = None
    if sampler is not None:
        th = threading.Thread(target=poll, daemon=True)
        th.start()
        time.sleep(0.5)             # idle baseline
    t0 = time.perf_counter()
    proc = subprocess.run(cmd, capture_output=True, text=True)
    t1 = time.perf_counter()
    if th is not None:
        time.sleep(0.3)
        stop.set()
        th.join()
    sys.stdout.write(proc.stdout)
    sys.stderr.write(proc.stderr[-2000:])
    rec = {"cmd": " ".join(cmd), "rc": proc.returncode, "sampler": getattr(sampler, "name", None), "sampler_errors": errors,
           "power_cap_W": getattr(sampler, "cap_w", None), "wall_s": t1 - t0, "samples": len(trace)}
    bench = None
    for line in proc.stdout.splitlines():
        if line.startswith("{"):
            try:
                bench = json.loads(line)
            except Exception:
                pass
    if bench:
        rec["bench"] = {k: bench.get(k) for k in ("value", "unit", "steps", "ms_per_step")}
        rec["bench"]["kernel"] = bench.get("config", {}).get("kernel")
        rec["bench"]["workload"] = bench.get("config", {}).get("workload", "")[:60]
        rec["bench"]["pixel_iterations_per_step"] = bench.get("config", {}).get("pixel_iterations_per_step_per_gpu")
    pw = [(t, p, c, e) for t, p, c, e in trace if p is not None]
    if pw:
        idle = sorted(p for t, p, c, e in pw if t < t0)[: max(1, len([1 for t, *_ in pw if t < t0]))]
        idle_w = sum(idle) / len(idle) if idle else min(p for _, p, _, _ in pw)
        peak_w = max(p for _, p, _, _ in pw)
        thr = idle_w + 0.2 * (peak_w - idle_w)
        busy = [(t, p, c, e) for t, p, c, e in pw if p > thr and t0 <= t <= t1 + 0.3]
        rec.update({"idle_power_W": idle_w, "peak_power_W": peak_w, "busy_samples": len(busy)})
        if busy:
            ps = sorted(p for _, p, _, _ in busy)
            cs = sorted(c for _, _, c, _ in busy if c)
            rec["busy_power_W"] = {"mean": sum(ps) / len(ps), "p50": ps[len(ps) // 2], "p95": ps[int(len(ps) * 0.95)]}
            if cs:
                rec["busy_sclk_MHz"] = {"mean": sum(cs) / len(cs), "p50": cs[len(cs) // 2], "min": cs[0], "max": cs[-1]}
            if bench and bench.get("value"):
                rec["J_per_G_pixel_iteration"] = rec["busy_power_W"]["mean"] / bench["value"]
            span = busy[-1][0] - busy[0][0]
            rec["busy_span_s"] = span
            if busy[0][3] is not None and busy[-1][3] is not None and busy[-1][3] > busy[0][3]:
                rec["busy_energy_J_counter"] = busy[-1][3] - busy[0][3]
            # trapezoid over the power samples as a cross-check / fallback
            ej = sum(0.5 * (busy[i][1] + busy[i + 1][1]) * (busy[i + 1][0] - busy[i][0]) for i in range(len(busy) - 1))
            rec["busy_energy_J_integrated"] = ej
        step = max(1, len(trace) // 2000)
        rec["trace_t_power_W_sclk_MHz"] = [[round(t - t0, 4), p, c] for t, p, c, e in trace[::step]]
    with open(out_path, "w") as f:
        json.dump(rec, f)
    brief = {k: v for k, v in rec.items() if k != "trace_t_power_W_sclk_MHz"}
    print("[power_trace]", json.dumps(brief))


if __name__ == "__main__":
    main()
