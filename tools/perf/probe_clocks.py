"""Clock / power probe (GPU box): what the shader clock and the package power do while the matrix kernels run - is the "throttled pipe"
of DESIGN.md 10.9 (1.4 of 2.5 PFLOP/s on random fp16 operands, 140 of 157 TFLOP/s on random fp32 operands) a CLOCK that falls under a
power cap?  A sampler thread reads amdgpu's sysfs files (current sclk level of pp_dpm_sclk, hwmon power1_average / power1_input, the
package power cap) every ~20 ms while the main thread runs, for a few seconds each: the fp32 / fp16 matrix probes on constant and on
random operands, the headline frame at fp32 / f16x3 / f16, and an idle gap.  Prints min / mean / max per phase.
    python tools/perf/probe_clocks.py [seconds per phase]"""
import ctypes as C
import glob
import os
import re
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import _lib, configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def find_card():
    """The DRM node of the GPU this process computes on: the box exposes one node per GPU of the host, only one is ours - the one
    whose power moves when a matrix probe runs."""
    cards = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_input"))
             or glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_average"))]
    if len(cards) <= 1:
        return cards[0] if cards else None

    def power(d):
        for name in ("power1_average", "power1_input"):
            for f in glob.glob(os.path.join(d, "hwmon", "hwmon*", name)):
                t = read(f)
                if t and t.strip().isdigit():
                    return int(t) / 1e6
        return 0.0
    idle = {d: power(d) for d in cards}
    lib = _lib.load()
    x = torch.zeros(1, device="cuda")
    best = {d: 0.0 for d in cards}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        tf, ms = C.c_double(), C.c_double()
        lib.pr_probe_mfma_f32(20000, 1, C.byref(tf), C.byref(ms), None)
        for d in cards:
            best[d] = max(best[d], power(d) - idle[d])
    del x
    print("power rise per node under a probe:", {os.path.basename(os.path.dirname(d)): round(v) for d, v in best.items()})
    return max(cards, key=lambda d: best[d])



class Sampler(threading.Thread):
    def __init__(self, dev):
        super().__init__(daemon=True)
        self.dev = dev
        hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        self.hw = hw[0] if hw else None
        self.samples = []
        self.label = "idle"
        self.stop = False

    def run(self):
        while not self.stop:
            sclk = None
            text = read(os.path.join(self.dev, "pp_dpm_sclk"))
            if text:
                for line in text.splitlines():
                    if "*" in line:
                        m = re.search(r"(\d+)\s*Mhz", line, re.I)
                        if m:
                            sclk = int(m.group(1))
            power = None
            if self.hw:
                for name in ("power1_average", "power1_input"):
                    t = read(os.path.join(self.hw, name))
                    if t and t.strip().isdigit():
                        power = int(t) / 1e6
                        break
            freq = None
            if self.hw:
                t = read(os.path.join(self.hw, "freq1_input"))
                if t and t.strip().isdigit():
                    freq = int(t) / 1e6
            self.samples.append((self.label, sclk, power, freq))
            time.sleep(0.02)


def summarise(samples, label):
    rows = [s for s in samples if s[0] == label]
    out = [f"{label:28s} n={len(rows):4d}"]
    for idx, name, unit in ((1, "sclk(dpm)", "MHz"), (3, "freq1", "MHz"), (2, "power", "W")):
        vals = [r[idx] for r in rows if r[idx] is not None]
        if vals:
            out.append(f"{name} {min(vals):7.0f} / {sum(vals) / len(vals):7.0f} / {max(vals):7.0f} {unit}")
    return "   ".join(out)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    dev_path = find_card()
    print("sysfs device:", dev_path)
    if dev_path:
        hw = glob.glob(os.path.join(dev_path, "hwmon", "hwmon*"))
        if hw:
            for name in ("power1_cap", "power1_cap_max", "power1_cap_default"):
                t = read(os.path.join(hw[0], name))
                if t:
                    print(name, int(t) / 1e6, "W")
        print("pp_dpm_sclk:", (read(os.path.join(dev_path, "pp_dpm_sclk")) or "").replace("\n", " | "))
    sampler = Sampler(dev_path) if dev_path else None
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    cfg = configs.tennis_config(hierarchical=(64, 128))
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    model.frame_replay = None
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
    model.eval().to(dev)
    size = (256, 256)
    scene = bench.to_device(synthetic.tennis_scene(seed=1234, image_size=size), dev)

    def frame():
        with torch.no_grad():
            model(*bench.scene_args(scene, size), 0, False, mode="scene_encodings")
    for precision in ("fp32", "f16x3", "f16"):
        model.object_composer.precision = precision
        frame()
    torch.cuda.synchronize()
    if sampler:
        sampler.start()
    phases = []

    def phase(label, fn):
        if sampler:
            sampler.label = label
        t0 = time.perf_counter()
        n = 0
        rates = []
        while time.perf_counter() - t0 < seconds:
            r = fn()
            if r is not None:
                rates.append(r)
            n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        phases.append((label, n, dt, sum(rates) / len(rates) if rates else None))
        if sampler:
            sampler.label = "gap"
        time.sleep(1.0)

    def probe(name, iterations, rnd):
        def run():
            tf, ms = C.c_double(), C.c_double()
            _lib.check(getattr(lib, name)(iterations, rnd, C.byref(tf), C.byref(ms), None), name)
            return tf.value
        return run
    time.sleep(1.0)
    phase("probe fp32 constant", probe("pr_probe_mfma_f32", 100000, 0))
    phase("probe fp32 random", probe("pr_probe_mfma_f32", 100000, 1))
    phase("probe fp16 constant", probe("pr_probe_mfma_f16", 100000, 0))
    phase("probe fp16 random", probe("pr_probe_mfma_f16", 100000, 1))
    for precision in ("fp32", "f16x3", "f16"):
        model.object_composer.precision = precision

        def run():
            frame()
            torch.cuda.synchronize()
        phase(f"headline frame {precision}", run)
    if sampler:
        sampler.stop = True
        sampler.join()
    for label, n, dt, rate in phases:
        extra = f"  {rate:8.1f} TFLOP/s (probe)" if rate is not None else f"  {dt / n * 1e3:8.2f} ms per frame"
        print((summarise(sampler.samples, label) if sampler else label) + extra)
    if sampler:
        print(summarise(sampler.samples, "gap"))


if __name__ == "__main__":
    main()
