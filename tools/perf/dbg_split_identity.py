"""Bit-identity check of two builds of the library on one box (GPU): one split-precision training forward + backward of bench.py's
training leg (minecraft, 3 x 2880 rays, perturb, train-mode BatchNorm, precision f16x3) with the library named by PR_PERF_LIB (default:
the product build); writes a SHA-256 per output field / parameter gradient to the JSON file given as argv[1].

    python tools/perf/dbg_split_identity.py gpurun_out/id_base.json
    PR_PERF_LIB=build/variants/libplayrender_old.so python tools/perf/dbg_split_identity.py gpurun_out/id_old.json
    python tools/perf/dbg_split_identity.py --compare gpurun_out/id_base.json gpurun_out/id_old.json

Used for the v_fma_mix operand split: the new sequence must produce the SAME fp16 pairs as the old one."""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def run(path):
    import bench
    from playableenvironments_amd import _lib, configs, synthetic
    from playableenvironments_amd.environment_model import EnvironmentModel
    if os.environ.get("PR_PERF_LIB"):
        _lib.library_path = lambda: os.path.abspath(os.environ["PR_PERF_LIB"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.train().to(dev)
    size = (288, 512)
    sc = bench.to_device(synthetic.minecraft_scene(batch=3, seed=77, image_size=size), dev)
    for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
        sc[k].requires_grad_(True)
    comp = model.object_composer
    comp.precision = os.environ.get("PR_PERF_PRECISION", "f16x3")
    record = {"library": _lib.library_path(), "fields": {}, "gradients": {}, "inputs": {}}
    for rep in range(2):        # (two runs: what differs between them is atomic-order noise, not the build)
        torch.manual_seed(123)
        for p in comp.parameters():
            p.grad = None
        for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
            sc[k].grad = None
        out = model(*bench.scene_args(sc, size), 2880, True, 0, patch_size=48, patch_stride=[4, 8], mode="scene_encodings")
        loss = out["coarse"]["global"]["integrated_features"].square().mean()
        loss.backward()
        torch.cuda.synchronize()
        fields = {f"coarse.global.{k}": digest(v) for k, v in out["coarse"]["global"].items() if torch.is_tensor(v)}
        grads = {n: digest(p.grad) for n, p in comp.named_parameters() if p.grad is not None}
        ins = {k: digest(sc[k].grad) for k in ("object_rotation_parameters", "object_translation_parameters", "object_style",
                                               "object_deformation") if sc[k].grad is not None}
        if rep == 0:
            record["fields"], record["gradients"], record["inputs"] = fields, grads, ins
        else:
            record["unstable"] = sorted([k for k in fields if fields[k] != record["fields"][k]] +
                                        [k for k in grads if grads[k] != record["gradients"][k]] +
                                        [k for k in ins if ins[k] != record["inputs"][k]])
    record["loss"] = float(loss)
    with open(path, "w") as f:
        json.dump(record, f, indent=1)
    print(f"{record['library']}: {len(record['fields'])} fields, {len(record['gradients'])} parameter gradients, "
          f"{len(record['unstable'])} differ between two runs of the same build")


def compare(a, b):
    ra, rb = json.load(open(a)), json.load(open(b))
    noisy = set(ra["unstable"]) | set(rb["unstable"])
    worst = 0
    for group in ("fields", "gradients", "inputs"):
        keys = sorted(set(ra[group]) | set(rb[group]))
        same = [k for k in keys if ra[group].get(k) == rb[group].get(k)]
        diff = [k for k in keys if k not in same]
        stable_diff = [k for k in diff if k not in noisy]
        worst += len(stable_diff)
        print(f"{group}: {len(same)} / {len(keys)} bit-identical; differing and run-to-run stable: {stable_diff[:8]}"
              + (f" (+ {len(diff) - len(stable_diff)} that also differ between runs of one build)" if len(diff) != len(stable_diff) else ""))
    print("IDENTICAL" if worst == 0 else "DIFFERENT")
    return worst


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
    run(sys.argv[1])
