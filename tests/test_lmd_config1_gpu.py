"""BASELINE config[0] on MI355X against THE REFERENCE ITSELF at full width.

tests/golden/run_lmd_sd15_config1.npz was recorded from the reference's OWN, unmodified `generation/lmd.run` on the
full-width SD1.5 network (859 M parameters, seeded synthetic weights), fp32, CPU, 10 DDIM steps, every default argument
(oracle/make_golden_config1.py): the latents entering every denoising step of the two per-box generations and of the
overall generation, the guidance iteration counts and losses, the reference maps of the transfer term, the composed
latents, the final latents.  Here the HIP engine (fp16 compute, INTEGRATION.md "Precision contract") replays

  * every step of every generation TEACHER-FORCED from the reference's latents of that step — guidance iterations
    (per-box: the box energy; overall: + reference-attention transfer), CFG pass, DDIM update, frozen blend;
  * the whole `run()` free-running through `lgd_amd.pipeline.lmd_generate` (= the plugin's body) from the same seeds.

Tolerances: teacher-forced steps at 3x what MI355X measured against the reference's own states (overall generation: every
guidance loss <= 2.3e-4, latents 8.0e-3 after step 0, <= 8.3e-4 after every later step; per-box generations 6.7e-3 / <= 2.4e-3);
the free-running run — 3 x 35 guidance iterations behind the energy's top-k selection — lands on the reference's composed
latents within 1.6e-2 and on its final latents within 7.6e-3 (rel-L2)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "run_lmd_sd15_config1.npz")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lgd_amd  # noqa: E402,F401
from lgd_amd import weights  # noqa: E402
from lgd_amd.sampler import LMDSampler  # noqa: E402
from lgd_amd.scheduler import DDIMScheduler  # noqa: E402
from lgd_amd.unet import UNetEngine  # noqa: E402
from conftest import gate  # noqa: E402

OBJ_KEY = ("down", 2, 1, 0)
T = 10
LIM_MAP0, LIM_MAP = 1.5e-1, 2.5e-2        # saved maps, rel-L2, 3x the measured worst key: step 0 4.9e-2 (others 1.0-1.9e-2), steps 1-9 8.4e-3
_S = {}


def relerr(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


def setup(dev):
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/run_lmd_sd15_config1.npz not generated (oracle/make_golden_config1.py, build container)")
    if not _S:
        cfg = weights.CONFIGS["sd15"]
        _S.update(cfg=cfg, g=np.load(GOLD), eng=UNetEngine(cfg, dev, weights.synth_state_dict(cfg, 0)))
    return _S


def _so_guidance(g, i):
    kw = json.loads(str(g["so_guidance_kwargs"]))[i]
    kw.pop("verbose", None)
    kw["guidance_attn_keys"] = [tuple(k) for k in kw["guidance_attn_keys"]]
    return dict(bboxes=json.loads(str(g[f"g{i}_bboxes"])), object_positions=json.loads(str(g[f"g{i}_object_positions"])), **kw)


def _check_steps(sm, tag, starts, final, iters, losses, loss_scale, ehs, lim_guided, lim_later, **kw):
    n0 = 0
    worst = 0.0
    for s in range(T):
        tr = []
        out = sm.denoise(torch.from_numpy(starts[s]) if "hist" not in kw else kw["hist"](s), ehs, T, first_step=s, n_steps=1,
                         trace=tr, **{k: v for k, v in kw.items() if k not in ("hist", "saved_ref", "saved_keys", "map_err")})
        want = starts[s + 1] if s < T - 1 else final
        assert out["guidance_iters"] == int(iters[s]), (tag, s, out["guidance_iters"], int(iters[s]))
        if kw.get("saved_ref") is not None:
            # the maps this step SAVED (pipelines.py:129-247 return_saved_cross_attn: condition half, word token) against the
            # reference's own, before lmd.run aligns them: {key: [T, Bp = 1, heads, HW, 1]} (row = the absolute step index;
            # a one-step run fills row s only) vs [T, heads, HW]
            for ki, k_ in enumerate(kw["saved_keys"]):
                got_m = out["saved"][k_][s, 0, :, :, 0].float().cpu()
                ref_m = torch.from_numpy(kw["saved_ref"][ki][s].astype(np.float32))
                # step 0 (four guidance iterations from noise: the chaotic one) apart from the later steps
                prev = kw["map_err"].get((k_, s == 0), (0.0, 0.0))
                kw["map_err"][(k_, s == 0)] = (max(prev[0], float((got_m - ref_m).abs().max() / ref_m.abs().max().clamp_min(1e-12))),
                                               max(prev[1], rel_l2(got_m, ref_m)))
        if iters[s]:
            got = np.array([x["loss"] for x in tr]) / loss_scale
            ref = losses[n0:n0 + int(iters[s])]
            gate(f"[config 1, {tag}] step {s}: {int(iters[s])} guidance losses, max rel. error", float(np.abs(got - ref).max() / np.abs(ref).max()), 1e-3)
        n0 += int(iters[s])
        e = relerr(out["latents_all"][s + 1], want)
        worst = max(worst, e)
        # step 0 starts from noise with the largest updates (4 iterations x sqrt(1 - abar) ~ 1): its own limit
        gate(f"[config 1, {tag}] step {s} teacher-forced ({int(iters[s])} guidance iterations): latents relerr", e,
             (lim_guided if s == 0 else lim_later) if iters[s] else 1e-3)
    return worst


def test_config1_per_box_generations_teacher_forced_vs_the_reference_run(dev):
    """generate_semantic_guidance of lmd.run's per-box stage (lmd.py:99-149 -> pipelines.py:129-247): guidance on the
    centred box with the energy's default weights, map saving for the word token."""
    s = setup(dev)
    g = s["g"]
    sm = LMDSampler(s["eng"], DDIMScheduler())
    mpath = os.path.join(ROOT, "tests", "golden", "run_lmd_sd15_config1_maps.npz")
    gm = np.load(mpath) if os.path.exists(mpath) else None
    for i in (0, 1):
        gd = _so_guidance(g, i)
        ehs = torch.from_numpy(g[f"g{i}_text_embeddings"])
        assert relerr(g[f"g{i}_starts"][0], g[f"g{i}_latents_in"]) == 0.0
        keys = [OBJ_KEY, *gd["guidance_attn_keys"]]
        if gm is not None:
            assert [tuple(k) for k in json.loads(str(gm[f"so{i}_saved_keys"]))] == keys
        errs = {}
        _check_steps(sm, f"per-box generation {i}", g[f"g{i}_starts"], g[f"g{i}_final"], g[f"g{i}_iters"], g[f"g{i}_losses"],
                     gd["loss_scale"], ehs, 2e-2, 7e-3, guidance=gd,       # measured: step 0 6.7e-3 / 6.3e-3; later <= 2.4e-3
                     saved_cross_attn_keys=keys,
                     return_cond_ca_only=True, return_token_ca_only=gd["object_positions"][0][-1],
                     saved_ref=[gm[f"so{i}_saved_k{ki}"] for ki in range(len(keys))] if gm is not None else None,
                     saved_keys=keys, map_err=errs)
        print(f"[config 1, per-box generation {i}] saved word-token maps (max-norm / rel-L2): "
              + ", ".join(f"{k_} {'step 0' if first else 'steps 1-9'}: {e[0]:.3e} / {e[1]:.3e}" for (k_, first), e in errs.items()))
        for (k_, first), (e_max, e_l2) in errs.items():
            # teacher-forced per step, so each map sits behind that step's guidance iterations only.  The reference's maps were
            # stored as fp16 (2^-11 relative), the engine's fp16 softmax on top.  The criterion is rel-L2 over the map: the word
            # token's probabilities reach 1.0 on single positions in this random-weight network and the max-norm is taken
            # exactly there (printed above, not gated).  First valid measurement (round 6, after the row-index fix).
            tag = "step 0, four guidance iterations" if first else "steps 1-9"
            gate(f"[config 1, per-box generation {i}] saved word-token maps at {k_} vs the reference's ({tag}, rel-L2)", e_l2, LIM_MAP0 if first else LIM_MAP)


def test_config1_overall_generation_teacher_forced_vs_the_reference_run(dev):
    """generate_partial_frozen of lmd.run's overall stage (lmd.py:530-542 -> pipelines.py:541-599): guidance with the
    reference-attention transfer on the REFERENCE's own (aligned) per-box maps, frozen blend on its composed latents."""
    s = setup(dev)
    g = s["g"]
    sm = LMDSampler(s["eng"], DDIMScheduler())
    kw = json.loads(str(g["ov_guidance_kwargs"]))
    kw.pop("verbose", None)
    keys = kw["guidance_attn_keys"] = [tuple(k) for k in kw["guidance_attn_keys"]]
    bboxes, pos = json.loads(str(g["g2_bboxes"])), json.loads(str(g["g2_object_positions"]))
    hw = sm.map_hw(64)
    max_hw = max(hw[k] for k in keys)
    ref_maps = torch.zeros((T, len(bboxes), len(keys), 8, max_hw))
    for o in range(len(bboxes)):
        for ki, k in enumerate(keys):
            ref_maps[:, o, ki, :, :hw[k]] = torch.from_numpy(g[f"ov_ref_o{o}_k{ki}"])
    gd = dict(bboxes=bboxes, object_positions=pos, ref_maps=ref_maps.to(dev), **kw)
    ehs = torch.from_numpy(g["g2_text_embeddings"])
    hist_ref = torch.from_numpy(g["ov_latents_all"])                         # (T+1, 1, 4, 64, 64): the composed latents
    assert relerr(g["composed"], g["ov_latents_all"]) == 0.0
    fm = torch.from_numpy(g["ov_frozen_mask"])
    fs = int(g["ov_frozen_steps"])

    def hist(step):                                                          # state before `step` + the blend rows behind it
        h = hist_ref.clone()
        h[0] = torch.from_numpy(g["g2_starts"][step])
        return h
    # measured on MI355X: losses <= 2.3e-4; latents 8.0e-3 after step 0 (4 iterations from noise), <= 8.3e-4 after every later step
    _check_steps(sm, "overall generation", g["g2_starts"], g["g2_final"], g["g2_iters"], g["g2_losses"], kw["loss_scale"], ehs,
                 2.5e-2, 2.5e-3, guidance=gd, frozen_steps=fs, frozen_mask=fm, hist=hist)


def test_config1_whole_run_free_running_vs_the_reference_run(dev):
    """The plugin body (`lmd_generate` with lmd.run's defaults) from the same seeds and embeddings: composition inputs are
    bit-exact host work, the per-box histories and the final latents carry the free-running fp16 error."""
    from lgd_amd.pipeline import CachedLayout, convert_box, lmd_generate
    s = setup(dev)
    g = s["g"]
    spec, kw = json.loads(str(g["spec"])), json.loads(str(g["kwargs"]))
    named = sorted(((n, convert_box(b)) for n, b in spec["gen_boxes"]), key=lambda nb: nb[0])      # parse.convert_spec's order
    lay = CachedLayout(boxes=[tuple(b) for _, b in named],
                       so_uncond=torch.from_numpy(g["g0_text_embeddings"][0:1]),
                       so_cond=torch.cat([torch.from_numpy(g[f"g{i}_text_embeddings"][1:2]) for i in (0, 1)]),
                       so_object_positions=[json.loads(str(g[f"g{i}_object_positions"]))[0] for i in (0, 1)],
                       so_word_token_index=[json.loads(str(g[f"g{i}_object_positions"]))[0][-1] for i in (0, 1)],
                       overall_uncond=torch.from_numpy(g["g2_text_embeddings"][0:1]),
                       overall_cond=torch.from_numpy(g["g2_text_embeddings"][1:2]), overall_groups=[[0], [1]],
                       overall_object_positions=json.loads(str(g["g2_object_positions"])),
                       overall_word_token_indices=json.loads(str(g["ov_guidance_kwargs"]))["word_token_indices"],
                       phrase_embeddings=torch.zeros(2, 768), bg_seed=kw["bg_seed"], fg_seed_start=kw["fg_seed_start"])
    sm = LMDSampler(s["eng"], DDIMScheduler())
    out = lmd_generate(sm, lay, num_inference_steps=T, frozen_step_ratio=0.5, so_center_box=True, align_with_overall_bboxes=True,
                       so_horizontal_center_only=False, horizontal_shift_only=False, decode=False)
    assert torch.equal(out["fg_idx"].cpu(), torch.from_numpy(g["fg_idx"]))
    assert out["so_guidance_iters"] == [int(g["g0_iters"].sum()), int(g["g1_iters"].sum())]
    assert out["guidance_iters"] == int(g["g2_iters"].sum())
    # measured on MI355X against the reference's own full-width fp32 run: 1.6e-2 / 1.0e-2 / 7.6e-3
    gate("[config 1, run] composed latents (two free-running guided per-box generations)", relerr(out["composed"], g["composed"]), 5e-2)
    gate("[config 1, run] composed latents rel-L2", rel_l2(out["composed"], g["composed"]), 3e-2)
    gate("[config 1, run] final latents rel-L2 (free-running, 3 x 35 guidance iterations)", rel_l2(out["latents"], g["g2_final"]), 2.5e-2)
