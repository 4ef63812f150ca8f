"""Check run by tests/test_gpu.py (own process: a runtime defect must not take the test session down): the same training
iterations as eager launches and as a recorded HIP graph (frame_graph.GraphedStep) follow the same loss trajectory and end with
the same parameters, with host synchronisations and eager kernels between bursts of replays - the pattern that kills the memset
nodes of a graph on ROCm 7.0.2 unless the runtime switch of frame_graph.GRAPH_RUNTIME_SWITCH is set - and the recorded call draws
fresh noise per replay."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from playableenvironments_amd import configs, synthetic  # noqa: E402
from playableenvironments_amd.frame_graph import GraphedStep, graph_runtime_is_safe  # noqa: E402
from playableenvironments_amd.object_composer import ObjectComposer  # noqa: E402
from tests.helpers import composer_inputs, grid_pixels  # noqa: E402


def run(recorded: bool, perturb: bool, steps: int = 12):
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    comp = ObjectComposer(cfg)
    synthetic.randomize_module_state(comp, seed=0, step=20000, alpha_bias=1.0, bender_scale=1e4)
    comp = comp.cuda().train()
    scene = synthetic.minecraft_scene(batch=2, seed=5)
    h, w = scene["image_size"]
    inputs = [v.cuda() for v in composer_inputs(cfg, scene, pixels=grid_pixels(h, w, 24))]
    o, d, n, w2o, sty, dfm, ins = inputs
    sty.requires_grad_(True)
    params = list(comp.parameters())
    # plain SGD: the update is proportional to the gradient (Adam's m / sqrt(v) turns the last-bit differences of atomically
    # accumulated gradients into full-size steps wherever a gradient is close to zero)
    opt = torch.optim.SGD(params, lr=1e-3)
    seeds = []

    def step():
        opt.zero_grad(set_to_none=True)
        out = comp(o, d, n, w2o, sty, dfm, ins, perturb)
        loss = out["coarse"]["global"]["integrated_features"].square().mean() + out["coarse"]["object_2"]["opacity"].mean()
        loss.backward()
        opt.step()
        return loss

    fn = step
    if recorded:
        comp.noise_seed_source = "device"
        graph = GraphedStep(step, warmup=1, modules=comp)
        fn = graph.replay
    else:
        step()      # the recorded run's warm-up iteration (recording itself executes nothing)
    losses = []
    for i in range(steps):
        losses.append(fn().detach().clone())
        if recorded and perturb:
            seeds.append(comp.last_noise_seed.clone())
        if i % 5 == 4:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    result = ([p.detach().clone() for p in params], [float(v) for v in torch.stack(losses)], seeds, [n for n, _ in comp.named_parameters()])
    if recorded and not perturb:
        # an eager evaluation render BETWEEN replays must see the weights the replays left on the device (the tensors' version
        # counters did not move: GraphedStep drops the composer's packed copies after every replay), and the annealing weights a
        # recorded step baked in make replay() refuse once set_step has moved them
        def eval_render():
            comp.eval()
            with torch.no_grad():
                feats = comp(o, d, n, w2o, sty.detach(), dfm, ins, False)["coarse"]["global"]["integrated_features"].clone()
            comp.train()
            return feats
        first = eval_render()
        for _ in range(3):
            fn()
        second = eval_render()
        comp._drop_device_caches()
        fresh = eval_render()
        assert not torch.equal(first, second), "an eager render after further replays returned the stale packed weights"
        assert torch.equal(second, fresh), "eager render between replays differs from a render with freshly packed weights"
        comp.set_step(20000)
        fn()                                   # same annealing weights: fine
        comp.set_step(30000)
        try:
            fn()
        except RuntimeError as e:
            assert "annealing" in str(e)
        else:
            raise AssertionError("replay() accepted a step whose annealing weights differ from the recorded ones")
        print("eager-between-replays and annealing checks ok")
    return result


def main():
    assert graph_runtime_is_safe(), "run with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
    # without noise both runs are the same arithmetic up to the order of atomic additions; the networks amplify such last-bit
    # differences from step to step (ReLU / box decisions flip), so the yardstick is a SECOND eager run
    def distance(pa, pb):
        worst = 0.0
        for a, b in zip(pa, pb):
            assert torch.isfinite(b).all()
            worst = max(worst, float((a - b).abs().max() / (a.abs().max() + 1e-12)))
        return worst
    p_eager, l_eager, _, names = run(False, False)
    p_again, l_again, _, _ = run(False, False)
    p_graph, l_graph, _, _ = run(True, False)
    for nm, a, b, c in zip(names, p_eager, p_again, p_graph):
        da = float((a - b).abs().max() / (a.abs().max() + 1e-12)); dg = float((a - c).abs().max() / (a.abs().max() + 1e-12))
        if dg > 1e-4: print(f'{nm}: eager/eager {da:.2e} eager/recorded {dg:.2e} max|p| {float(a.abs().max()):.3e}')
    noise_floor = distance(p_eager, p_again)
    worst = distance(p_eager, p_graph)
    print(f"eager vs eager {noise_floor:.3e}, eager vs recorded {worst:.3e}; losses {l_eager[-1]:.6f} {l_again[-1]:.6f} {l_graph[-1]:.6f}")
    for i in range(len(l_eager)):
        print(i, f"{l_eager[i]:.7f} {l_again[i]:.7f} {l_graph[i]:.7f}")
    assert worst <= max(10.0 * noise_floor, 2e-2), f"parameters after 12 iterations: recorded differs by {worst:.3e}, two eager runs by {noise_floor:.3e}"
    # the loss trajectory is the robust yardstick (stale zero-fills freeze it or blow it up)
    for a, b in zip(l_eager, l_graph):
        assert abs(a - b) <= 2e-4 * abs(a), (l_eager, l_graph)
    assert l_graph[-1] < l_graph[0]
    # with noise: every replay draws its own seed word, the loss stays finite and keeps changing
    _, l_noise, seeds, _ = run(True, True)
    values = sorted(int(s) for s in seeds)
    assert len(set(values)) == len(values), "the recorded call re-used a noise seed"
    assert all(v == v and abs(v) < 1e6 for v in l_noise) and len(set(l_noise)) > 6
    print(f"GRAPH STEP OK worst relative parameter difference {worst:.2e}")


if __name__ == "__main__":
    main()
