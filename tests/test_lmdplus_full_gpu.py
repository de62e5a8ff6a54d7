"""The HEADLINE configuration (BASELINE config[1]'s method, LMD+ on the full-width sd14_gligen network) against THE
REFERENCE ITSELF.

tests/golden/run_lmd_plus_sd14gligen_full.npz was recorded from the reference's OWN, unmodified `generation/lmd_plus.run`
(lmd_plus.py:193-520) on the full-width SD1.4 + GLIGEN network (seeded synthetic weights), fp32, CPU, 512 x 512, 20 DDIM
steps, every default argument (oracle/make_golden_lmdplus_full.py; profiles/r06_config2_reference_cpu.json holds its
wall-clock): per-box histories, composed latents, foreground indices, the latents entering every step of the overall
generation, the guidance iteration count of every step, every guidance loss, the final latents.  The HIP engine (fp16
compute = the reference's own autocast arithmetic for LMD+, lmd_plus.py:226,336) replays

  * every step of the overall generation TEACHER-FORCED from the reference's latents of that step (guidance iterations with
    the reference-attention transfer, GLIGEN fuser on / off by the scheduled-sampling beta, CFG pass, DDIM update, frozen
    blend) through the plugin body's own hook (`overall_first_step / overall_n_steps / overall_start`);
  * the whole `run()` free-running through `lgd_amd.pipeline.lmd_plus_generate` from the same seeds and embeddings.

[ext] caveat: ResnetBlock2D / Timesteps / DDIMScheduler of the reference run come from oracle/stubs/diffusers
(diffusers itself is absent) — parity is unpinned at that boundary, here as everywhere."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "run_lmd_plus_sd14gligen_full.npz")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import gate  # noqa: E402
from fake_text import FakeTextEncoder, FakeTokenizer  # noqa: E402

T = 20
_S = {}


def relerr(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


def setup(dev):
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/run_lmd_plus_sd14gligen_full.npz not generated (oracle/make_golden_lmdplus_full.py, build container)")
    if not _S:
        sys.path.insert(0, os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin"))
        sys.modules.pop("inflect", None)
        sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))       # `inflect` stand-in, as in the golden run
        import lgd_amd  # noqa: F401
        from lgd_amd import weights
        import models
        cfg = weights.CONFIGS["sd14_gligen"]
        keep = models.model_dict
        md = models.build_model_dict(cfg, weights.synth_state_dict(cfg, 0), vae=None, tokenizer=FakeTokenizer(),
                                     text_encoder=FakeTextEncoder(cfg.cross_attention_dim))
        models.model_dict = md
        try:
            from generation._common import DEFAULT_OVERALL_NEGATIVE_PROMPT, DEFAULT_SO_NEGATIVE_PROMPT, build_layout
            g = np.load(GOLD)
            spec, kw = json.loads(str(g["spec"])), json.loads(str(g["kwargs"]))
            lay = build_layout(spec, kw["bg_seed"], kw["fg_seed_start"], DEFAULT_SO_NEGATIVE_PROMPT, DEFAULT_OVERALL_NEGATIVE_PROMPT, 512, 512)
        finally:
            models.model_dict = keep
            sys.path.remove(os.path.join(ROOT, "oracle", "stubs"))
            sys.modules.pop("inflect", None)
        _S.update(g=g, lay=lay, sm=md.sampler, kw=kw)
    return _S


def test_golden_is_the_default_configuration(dev):
    s = setup(dev)
    g = s["g"]
    sg = json.loads(str(g["ov_guidance_kwargs"]))
    sc = json.loads(str(g["ov_scalars"]))
    assert s["kw"]["num_inference_steps"] == T and sc == dict(gligen_scheduled_sampling_beta=0.4, frozen_steps=10, guidance_scale=7.5)
    assert sg["max_index_step"] == 30 and sg["loss_threshold"] == 5.0 and sg["ref_ca_loss_weight"] == 2.0 and bool(g["ov_has_ref"])
    assert list(g["ov_iters"]) == sg["max_iter"][:T] and len(g["ov_losses"]) == int(g["ov_iters"].sum()) == 55


def test_overall_generation_teacher_forced_vs_the_reference_run(dev):
    """pipelines.generate_gligen with semantic guidance (lmd_plus.py:496-511 -> pipelines.py:324-520): one step at a time from
    the reference's own state at the start of that step; the per-box stage upstream (histories for the frozen blend,
    reference maps of the transfer term) is this engine's own."""
    from lgd_amd.pipeline import lmd_plus_generate
    s = setup(dev)
    g, sm, lay = s["g"], s["sm"], s["lay"]
    starts, iters = g["ov_starts"], g["ov_iters"]
    worst = {}
    for i in range(T):
        out = lmd_plus_generate(sm, lay, num_inference_steps=T, decode=False, overall_first_step=i, overall_n_steps=1,
                                overall_start=[starts[i]])
        want = starts[i + 1] if i < T - 1 else g["final_latents"]
        assert out["guidance_iters"] == int(iters[i]), (i, out["guidance_iters"], int(iters[i]))
        e = relerr(out["latents"], want)
        kind = "guided+frozen" if i < 10 else "guided"
        worst[kind] = max(worst.get(kind, 0.0), e)
        # limits = 3x what MI355X measured: step 0 1.5e-2 (4 iterations from noise), step 1 5.3e-3, steps 2-9 <= 1.95e-3 (they
        # blend in the composed latents of this engine's own per-box stage, 1.8e-3 off the golden's), steps 10-19 <= 1.6e-4
        gate(f"[config 2 full width] overall step {i} teacher-forced ({int(iters[i])} guidance iterations): latents relerr", e,
             4.5e-2 if i == 0 else 1.6e-2 if i == 1 else 6e-3 if i < 10 else 5e-4)
    print("worst teacher-forced relerr:", worst)


def test_whole_run_free_running_vs_the_reference_run(dev):
    """The plugin body from the same seeds and embeddings: unguided per-box GLIGEN generations (histories), composition
    (bit-exact host work on those histories), 55 guidance iterations in the overall generation."""
    from lgd_amd.pipeline import lmd_plus_generate
    s = setup(dev)
    g, sm, lay = s["g"], s["sm"], s["lay"]
    out = lmd_plus_generate(sm, lay, num_inference_steps=T, decode=False)
    assert torch.equal(out["fg_idx"].cpu(), torch.from_numpy(g["fg_idx"]))
    assert out["guidance_iters"] == int(g["ov_iters"].sum())
    for i in (0, 1):
        gate(f"[config 2 full width, run] per-box history {i} (20 unguided GLIGEN steps, free-running)",
             relerr(out["so_latents_all"][i], g[f"so{i}_latents_all"]), 6.5e-3)               # measured 2.1e-3 / 1.6e-3
    # measured on MI355X against the reference's own full-width fp32 run: 1.8e-3 / 1.8e-3 / 8.4e-3
    gate("[config 2 full width, run] composed latents", relerr(out["composed"], g["composed"]), 5.5e-3)
    gate("[config 2 full width, run] composed latents rel-L2", rel_l2(out["composed"], g["composed"]), 5.5e-3)
    gate("[config 2 full width, run] final latents rel-L2 (free-running, 55 guidance iterations)", rel_l2(out["latents"], g["final_latents"]), 2.6e-2)
