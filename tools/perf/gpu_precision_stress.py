"""Manual report (GPU box): the split-precision kernel on a network whose function is unchanged but whose weights are badly
scaled - one backbone layer multiplied by `shrink`, the next layer's matching inputs by 1 / shrink (ReLU is positively
homogeneous).  The unscaled fp16 residuals have an ABSOLUTE floor of 3e-8 per operand, so a uniformly tiny layer loses
relative accuracy; this prints how much."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import render_oracle as ro
from playableenvironments_amd import ObjectComposer, configs, synthetic
from tests.helpers import composer_inputs, grid_pixels

cfg, scene = configs.tennis_config(), synthetic.tennis_scene(seed=7)
inputs = composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 24))
for shrink in (1.0, 1e-1, 1e-2, 1e-3):
    torch.manual_seed(0)
    comp = ObjectComposer(cfg)
    synthetic.randomize_module_state(comp, seed=0, step=20000, alpha_bias=3.0, bender_scale=1e4)
    comp.eval()
    with torch.no_grad():
        for model in comp.object_models_coarse:
            layers = model.nerf_model.backbone_layers
            layers[1].weight.mul_(shrink)
            layers[1].bias.mul_(shrink)
            layers[2].weight[:, :layers[1].out_features].div_(shrink)
    sd = {k: v.detach().clone() for k, v in comp.state_dict().items()}
    with torch.no_grad():
        want = ro.composer_forward(cfg, sd, *inputs, False, stable_merge=True)
        a = want["coarse"]["global"]["integrated_features"]
        comp = comp.cuda()
        for precision in ("fp32", "f16x3"):
            comp.precision = precision
            b = comp(*[t.cuda() for t in inputs], False)["coarse"]["global"]["integrated_features"].cpu()
            d = (a - b).abs()
            ok = bool(((d <= 1e-5 + 1e-4 * a.abs()).all()))
            print(f"shrink {shrink:7.0e} {precision:6s} max |diff| {float(d.max()):.3e}  mean |diff| {float(d.mean()):.3e}  within rtol 1e-4 / atol 1e-5: {ok}")
