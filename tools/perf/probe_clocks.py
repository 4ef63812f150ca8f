"""Clock / power probe (GPU box): what the shader clock and the package power do while the matrix kernels run - is the "throttled pipe"
of DESIGN.md 10.9 (1.4 of 2.5 PFLOP/s on random fp16 operands, 140 of 157 TFLOP/s on random fp32 operands) a CLOCK that falls under a
power cap?  A sampler thread reads amdgpu's sysfs files (current sclk level of pp_dpm_sclk, hwmon power1_average / power1_input, the
package power cap) every ~20 ms while the main thread runs, for a few seconds each: the fp32 / fp16 matrix probes on constant and on
random operands, the headline frame at fp32 / f16x3 / f16, and an idle gap.  Prints min / mean / max per phase.
    python tools/perf/probe_clocks.py [seconds per phase]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import gpu_telemetry  # noqa: E402
from playableenvironments_amd import _lib, configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402


def summarise(t, label):
    d = t.summary(label)
    parts = [f"{label:28s} n={d.get('samples', 0):4d}"]
    if "sclk_mhz" in d:
        parts.append(f"sclk {d['sclk_mhz_min']:7.0f} / {d['sclk_mhz']:7.0f} / {d['sclk_mhz_max']:7.0f} MHz")
    if "power_w" in d:
        parts.append(f"power {d['power_w_min']:7.0f} / {d['power_w']:7.0f} / {d['power_w_max']:7.0f} W")
    return "   ".join(parts)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib = _lib.load()

    def load():
        tf, ms = C.c_double(), C.c_double()
        lib.pr_probe_mfma_f32(20000, 1, C.byref(tf), C.byref(ms), None)
    props = torch.cuda.get_device_properties(dev)
    print(f"device 0: {props.name} at PCI {props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0")
    by_address = gpu_telemetry.card_of_pci_address(props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    by_power = gpu_telemetry.find_card(load)
    print("DRM node by PCI address:", by_address, " by power rise:", by_power)
    dev_path = by_address or by_power
    print("sysfs device:", dev_path, " power cap:", gpu_telemetry.power_cap_w(dev_path) if dev_path else None, "W")
    sampler = gpu_telemetry.Telemetry(dev_path) if dev_path else None
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
        if sampler:
            sampler.label = None

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
        sampler.finish()
    for label, n, dt, rate in phases:
        extra = f"  {rate:8.1f} TFLOP/s (probe)" if rate is not None else f"  {dt / n * 1e3:8.2f} ms per frame"
        print((summarise(sampler, label) if sampler else label) + extra)
    if sampler:
        print(summarise(sampler, "gap"))


if __name__ == "__main__":
    main()
