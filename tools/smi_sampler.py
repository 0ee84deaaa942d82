"""Shader clock and socket power of GPU 0 sampled on a background thread while a timed region runs (bench.py's `roofline.clock_mhz` /
`frac_at_clock`, profiles/r05_power.txt).  Read-only: amdsmi's metrics table first, `rocm-smi --json` as the fallback.  Nothing here
touches a GPU setting, and nothing of it is on the product path.

  python tools/smi_sampler.py            prints what the box offers (one metrics sample, every key that looks like a clock or a power)
"""
from __future__ import annotations

import json
import subprocess
import threading
import time


def _num(x):
    try:
        v = float(x)
        return v if v == v and 0 < v < 1e7 else None        # N/A markers are 0xFFFF / 65535 or strings
    except Exception:   # noqa: BLE001
        return None


class _AmdSmi:
    def __init__(self, index=0):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[index]
        self.read()                                           # raises if the metrics table is not readable here

    def read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clks = m.get("current_gfxclks")
        if isinstance(clks, (list, tuple)):
            clks = [c for c in (_num(c) for c in clks) if c is not None and c < 60000]
        clk = (sum(clks) / len(clks)) if clks else _num(m.get("current_gfxclk"))
        if clk is None:
            clk = _num(m.get("average_gfxclk_frequency"))
        pw = _num(m.get("current_socket_power"))
        if pw is None or pw >= 65535:
            pw = _num(m.get("average_socket_power"))
        if clk is None and pw is None:
            raise RuntimeError("metrics table has neither a shader clock nor a socket power")
        return clk, pw

    def describe(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        return {k: v for k, v in m.items() if any(s in k for s in ("clk", "power", "energy", "temperature_hotspot", "throttle", "activity"))}


class _RocmSmi:
    def __init__(self, index=0):
        self.index = index
        self.read()

    def read(self):
        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                             timeout=20).stdout
        d = next(iter(json.loads(out).values()))
        clk = pw = None
        for k, v in d.items():
            kl = k.lower()
            if "sclk" in kl and "(" in str(v):
                clk = _num(str(v).split("(")[1].split("M")[0])
            if "power" in kl and "socket" in kl or "average graphics package power" in kl:
                pw = _num(v) if _num(v) else pw
        if clk is None and pw is None:
            raise RuntimeError("rocm-smi printed neither sclk nor power")
        return clk, pw

    def describe(self):
        return {"source": "rocm-smi --showclocks --showpower --json"}


def physical_index(local_index=0):
    """Index of the GPU the process calls device `local_index`, after HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (integer lists only)."""
    import os
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                ids = [int(x) for x in v.split(",") if x.strip() != ""]
                return ids[local_index] if local_index < len(ids) else local_index
            except ValueError:
                return local_index
    return local_index


def open_source(index=0, in_process_only=False):
    """-> (reader, name) or (None, reason).  in_process_only: no subprocess-based reader (forking `rocm-smi` every period from a process
    that enqueues a host-bound step perturbs what it annotates)."""
    errs = []
    for cls in ((_AmdSmi,) if in_process_only else (_AmdSmi, _RocmSmi)):
        try:
            return cls(index), cls.__name__.strip("_").lower()
        except Exception as e:   # noqa: BLE001
            errs.append(f"{cls.__name__}: {type(e).__name__}: {e}")
    return None, "; ".join(errs)


class Sampler:
    """with Sampler() as s: <timed region>;  s.summary() -> {'clock_mhz': mean, 'power_w': mean, 'samples': n, 'source': ...} or
    {'source': None, 'error': ...} when the box exposes nothing."""

    def __init__(self, period_s=0.02, index=0, in_process_only=False):
        self.period, self.src, self.name = period_s, None, None
        self.src, self.name = open_source(physical_index(index), in_process_only=in_process_only)
        self.rows, self._stop, self._t = [], threading.Event(), None

    def __enter__(self):
        if self.src is not None:
            self._stop.clear()
            self.rows = []
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            try:
                self.rows.append((time.perf_counter(),) + tuple(self.src.read()))
            except Exception:   # noqa: BLE001
                pass
            self._stop.wait(self.period)

    def __exit__(self, *exc):
        if self._t is not None:
            self._stop.set()
            self._t.join(timeout=5)
        return False

    def summary(self):
        if self.src is None:
            return {"source": None, "error": self.name}
        clk = [r[1] for r in self.rows if r[1] is not None]
        pw = [r[2] for r in self.rows if r[2] is not None]
        mean = lambda v: round(sum(v) / len(v), 1) if v else None   # noqa: E731
        return {"source": self.name, "samples": len(self.rows), "clock_mhz": mean(clk), "clock_mhz_min": min(clk) if clk else None,
                "clock_mhz_max": max(clk) if clk else None, "power_w": mean(pw), "power_w_max": max(pw) if pw else None}


if __name__ == "__main__":
    src, name = open_source()
    print("source:", name)
    if src is not None:
        print(json.dumps(src.describe(), indent=1, default=str))
        t0 = time.perf_counter()
        for _ in range(20):
            src.read()
        print(f"one read: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms ->", src.read())
