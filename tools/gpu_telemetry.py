"""Shader clock and package power of the GPU this process computes on, sampled from amdgpu's sysfs files by a background thread:
what `bench.py` records beside the headline's roofline (is the measured region clock limited?) and what `tools/perf/probe_clocks.py`
prints per phase.  Nothing here touches the device: it reads `/sys/class/drm/card*/device/pp_dpm_sclk` and `hwmon*/power1_*`.
Every failure (no sysfs, unreadable files) degrades to "no telemetry" - measurement code must never fail the measured run."""
import glob
import os
import re
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _power_w(dev):
    for name in ("power1_average", "power1_input"):
        for f in glob.glob(os.path.join(dev, "hwmon", "hwmon*", name)):
            t = _read(f)
            if t and t.strip().isdigit():
                return int(t) / 1e6
    return None


def _sclk_mhz(dev):
    for f in glob.glob(os.path.join(dev, "hwmon", "hwmon*", "freq1_input")):
        t = _read(f)
        if t and t.strip().isdigit():
            return int(t) / 1e6
    text = _read(os.path.join(dev, "pp_dpm_sclk"))
    if text:
        for line in text.splitlines():
            if "*" in line:
                m = re.search(r"(\d+)\s*Mhz", line, re.I)
                if m:
                    return float(m.group(1))
    return None


def power_cap_w(dev):
    for f in glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_cap")):
        t = _read(f)
        if t and t.strip().isdigit():
            return int(t) / 1e6
    return None


def card_of_pci_address(domain: int, bus: int, device: int):
    """The DRM node whose PCI address is domain:bus:device.0 (torch.cuda.get_device_properties(i).pci_domain_id / pci_bus_id /
    pci_device_id) - exact, where the sysfs tree shows the addresses (it does on the pool's boxes)."""
    want = f"{domain:04x}:{bus:02x}:{device:02x}."
    try:
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            if os.path.basename(os.path.realpath(d)).lower().startswith(want) and _power_w(d) is not None:
                return d
    except Exception:
        pass
    return None


def find_card(load, seconds: float = 1.5, pci=None):
    """The DRM node of the GPU `load()` runs on: a box exposes one node per GPU of its host, one of them is ours.  `pci` = (domain,
    bus, device) of the device names it exactly; otherwise it is the node whose package power rises most while `load()` (a callable
    that keeps the GPU busy for a few milliseconds per call) runs - which a busy neighbour on the same host can fool."""
    if pci is not None:
        exact = card_of_pci_address(*pci)
        if exact is not None:
            return exact
    try:
        cards = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if _power_w(d) is not None]
        if len(cards) <= 1:
            return cards[0] if cards else None
        idle = {d: _power_w(d) or 0.0 for d in cards}
        rise = {d: 0.0 for d in cards}
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            load()
            for d in cards:
                rise[d] = max(rise[d], (_power_w(d) or 0.0) - idle[d])
        best = max(cards, key=lambda d: rise[d])
        return best if rise[best] > 50.0 else None
    except Exception:
        return None


class Telemetry(threading.Thread):
    """Samples (label, sclk MHz, package W) every `period` seconds until `finish()`; `label` is set by the measuring thread."""

    def __init__(self, dev, period: float = 0.02):
        super().__init__(daemon=True)
        self.dev, self.period = dev, period
        self.samples = []
        self.label = None
        self._stop_flag = False

    def run(self):
        while not self._stop_flag:
            label = self.label
            if label is not None:
                try:
                    self.samples.append((label, _sclk_mhz(self.dev), _power_w(self.dev)))
                except Exception:
                    pass
            time.sleep(self.period)

    def finish(self):
        self._stop_flag = True
        self.join(timeout=2.0)

    def summary(self, label):
        rows = [s for s in self.samples if s[0] == label]
        out = {"samples": len(rows)}
        for idx, key in ((1, "sclk_mhz"), (2, "power_w")):
            vals = [r[idx] for r in rows if r[idx] is not None]
            if vals:
                out[key] = round(sum(vals) / len(vals), 1)
                out[key + "_min"] = round(min(vals), 1)
                out[key + "_max"] = round(max(vals), 1)
        cap = power_cap_w(self.dev)
        if cap is not None:
            out["power_cap_w"] = cap
        return out
