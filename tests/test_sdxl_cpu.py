"""Host side and oracle of the SDXL-refiner row (no GPU): the restatement against what the reference CAN pin, the
Euler sampler's algebra against the golden-pinned DDIM scheduler, the pipeline's bookkeeping, the drop-in's surface."""
import ast
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lgd_amd  # noqa: E402,F401
from lgd_amd import weights  # noqa: E402
from lgd_amd.scheduler import DDIMScheduler, EulerDiscreteScheduler  # noqa: E402
import restate as R  # noqa: E402
import restate_sdxl as X  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def test_oracle_block_layout_vs_reference_golden():
    """unet_fwd_tiny_outer.npz was produced by the reference's own UNet2DConditionModel (oracle/make_golden_outer.py)."""
    g = np.load(os.path.join(GOLD, "unet_fwd_tiny_outer.npz"))
    cfg = weights.CONFIGS["tiny_outer"]
    with torch.no_grad():
        e = X.unet_forward_xl(weights.synth_state_dict(cfg, 0), cfg, torch.from_numpy(g["x"]), int(g["t"]), torch.from_numpy(g["ehs"]))
    assert float((e - torch.from_numpy(g["eps"])).abs().max()) < 2e-5


def test_oracle_xl_forward_is_the_sd_forward_without_the_extensions():
    cfg = weights.CONFIGS["tiny_sd21"]
    sd = weights.synth_state_dict(cfg, 0)
    x, ehs = torch.randn(2, 4, 32, 32), torch.randn(2, 77, cfg.cross_attention_dim)
    with torch.no_grad():
        a = X.unet_forward_xl(sd, cfg, x, 501, ehs)
        b = R.unet_forward(sd, dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                                    attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups,
                                    norm_eps=cfg.norm_eps), x, 501, ehs)
    assert torch.equal(a, b)


def test_refiner_configuration():
    cfg = weights.CONFIGS["sdxl_refiner"]
    assert abs(weights.num_params(cfg) / 1e9 - 2.26) < 0.01            # the published size of the refiner UNet
    blocks = weights.unet_blocks(cfg)
    assert [len(b.attns) for b in blocks] == [0, 2, 2, 0, 1, 0, 3, 3, 0]
    assert all(a.depth == 4 and a.head_dim == 64 for b in blocks for a in b.attns)
    assert cfg.pooled_dim == 1280
    with pytest.raises(ValueError):
        cfg.to_ref_kwargs()                                              # the reference's UNet class cannot build it
    kw = weights.CONFIGS["tiny_outer"].to_ref_kwargs()
    assert kw["down_block_types"] == ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")


def test_euler_step_is_ddim_in_sigma_space():
    """x~ = x / sqrt(abar) turns the deterministic DDIM update into the Euler step on sigma = sqrt((1 - abar) / abar):
    the new sampler's algebra against the scheduler the reference goldens pin (tests/test_schedule.py)."""
    eu, dd = EulerDiscreteScheduler(), DDIMScheduler()
    eu.set_timesteps(50)
    dd.set_timesteps(50)
    assert [int(t) for t in eu.timesteps] == [int(t) for t in dd.timesteps]
    g = torch.Generator().manual_seed(0)
    for i in (0, 17, 35, 48):
        t, t_next = int(dd.timesteps[i]), int(dd.timesteps[i + 1])
        a, an = dd.alphas_cumprod[t].double(), dd.alphas_cumprod[t_next].double()
        x, eps = torch.randn(64, generator=g).double(), torch.randn(64, generator=g).double()
        x0 = (x - (1 - a).sqrt() * eps) / a.sqrt()
        ddim = an.sqrt() * x0 + (1 - an).sqrt() * eps
        euler = eu.step_host(eps, i, x / a.sqrt())
        assert float((euler * an.sqrt() - ddim).abs().max()) < 2e-5
        c0, c1, A, B, C, c_in = eu.multistep_rows()[i]
        xt = x / a.sqrt()
        assert float((A * xt + B * (c0 * xt + c1 * eps) - euler).abs().max()) < 1e-5
        assert abs(c_in - float(a.sqrt())) < 1e-6                          # scale_model_input returns the VP-space sample
    # last step lands on x0
    c0, c1, A, B, C, _ = eu.multistep_rows()[-1]
    assert A == 0.0 and B == 1.0 and C == 0.0


def test_img2img_bookkeeping_matches_the_oracle():
    eu = EulerDiscreteScheduler()
    eu.set_timesteps(50)
    o = X.EulerDiscrete()
    o.set_timesteps(50)
    assert torch.equal(eu.timesteps, o.timesteps) and torch.allclose(eu.sigmas, o.sigmas)
    assert eu.img2img_start(50, 0.3) == 35 and len(X.get_timesteps(o, 50, 0.3)[0]) == 15    # generate.py:52 default ratio
    assert eu.img2img_start(50, 0.5) == 25 and eu.img2img_start(50, 1.0) == 0
    from lgd_amd.sdxl import add_time_ids
    assert torch.equal(add_time_ids(1024, 1024), X.add_time_ids(1024, 1024))
    assert add_time_ids(1024, 1024).tolist() == [[1024, 1024, 0, 0, 2.5], [1024, 1024, 0, 0, 6.0]]


def test_dropin_module_surface_matches_the_reference():
    sys.path.insert(0, os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin"))
    import inspect
    import generation.sdxl_refinement as mod
    assert list(inspect.signature(mod.refine).parameters) == ["image", "spec", "refine_seed", "refinement_step_ratio"]
    assert inspect.signature(mod.refine).parameters["refinement_step_ratio"].default == 0.5
    assert inspect.signature(mod.init).parameters["offload_model"].default is True
    ref_file = "/root/reference/generation/sdxl_refinement.py"
    if os.path.exists(ref_file):                                         # build container only
        tree = ast.parse(open(ref_file).read())
        consts = {n.targets[0].id: ast.literal_eval(n.value) for n in tree.body
                  if isinstance(n, ast.Assign) and isinstance(n.value, ast.Constant)}
        assert mod.sdxl_negative_prompt == consts["sdxl_negative_prompt"]
        fns = {n.name: [a.arg for a in n.args.args] for n in tree.body if isinstance(n, ast.FunctionDef)}
        assert fns["refine"] == ["image", "spec", "refine_seed", "refinement_step_ratio"] and "init" in fns


def test_vae_encoder_host_algebra():
    """The two rewrites HipVAEEncoder relies on, in plain torch: quant_conv folded into conv_out, and Downsample2D's
    right / bottom padding as the odd positions of the stride-1 'same' convolution."""
    import torch.nn.functional as F
    from lgd_amd.vae import fold_pointwise_after, synth_aekl_state_dict
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 10, 10, generator=g)
    w, b = torch.randn(8, 16, 3, 3, generator=g), torch.randn(8, generator=g)
    w1, b1 = torch.randn(8, 8, 1, 1, generator=g), torch.randn(8, generator=g)
    wf, bf = fold_pointwise_after(w, b, w1, b1)
    ref = F.conv2d(F.conv2d(x, w, b, padding=1), w1, b1)
    assert float((F.conv2d(x, wf, bf, padding=1) - ref).abs().max()) < 1e-4
    wd = torch.randn(16, 16, 3, 3, generator=g)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), wd, stride=2)
    assert torch.allclose(F.conv2d(x, wd, padding=1)[:, :, 1::2, 1::2], ref, atol=1e-5)
    sd = synth_aekl_state_dict((32, 64), 1)
    assert sd["encoder.conv_out.weight"].shape == (8, 64, 3, 3) and sd["quant_conv.weight"].shape == (8, 8, 1, 1)
    assert "encoder.down_blocks.0.downsamplers.0.conv.weight" in sd and "encoder.down_blocks.1.downsamplers.0.conv.weight" not in sd
    assert sd["decoder.up_blocks.0.resnets.1.conv1.weight"].shape[0] == 64            # decoder mirrors the encoder widths
