"""ORACLE TEST INFRASTRUCTURE (build container only) — orchestration goldens.

Runs the reference's OWN, unmodified plugins `generation/lmd_plus.run` (lmd_plus.py:193-520) and
`generation/lmd.run` (lmd.py:215-551) on CPU through oracle/ref_harness.py, with
  * the reference UNet (tiny configs, seeded synthetic weights, fp32),
  * the whitespace fake tokenizer / table text encoder of tests/fake_text.py (no CLIP vocabulary here),
  * SAM replaced by the box mask (`utils.proportion_to_mask`), as SURVEY.md 8(d) prescribes for benchmarks,
  * recorder wrappers (no behaviour change) around latents.compose_latents_with_alignment,
    pipelines.generate_gligen / generate_partial_frozen and guidance.get_phrase_indices,
and writes tests/golden/run_lmd_plus_tiny.npz / run_lmd_tiny.npz: composed latents, foreground indices, the
final latents of the overall generation, per-box histories, and every get_phrase_indices call (prompt, phrases,
words -> positions), which also pins row G4 against the reference with the same fake tokenizer.

    python oracle/make_golden_runs.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness as H  # noqa: E402
from fake_text import FakeTextEncoder, FakeTokenizer  # noqa: E402

SPEC = dict(prompt="A realistic image of a white deer and a gray bear in an empty factory scene",
            gen_boxes=[("a white deer", [37, 88, 91, 117]), ("a gray bear", [157, 96, 94, 108])],
            bg_prompt="A realistic image of an empty factory scene", extra_neg_prompt="")
# a phrase that occurs twice (two boxes, pluralised overall phrase) and one absent from the prompt ("| phrase" suffix)
SPEC3 = dict(prompt="A photo of two apples on a table",
             gen_boxes=[("an apple", [20, 120, 80, 80]), ("an apple", [140, 110, 90, 90]), ("a wooden spoon", [60, 30, 120, 40])],
             bg_prompt="A photo of a table", extra_neg_prompt="cartoon")


def build(cfg_name):
    H.setup()
    import lgd_amd  # noqa: F401
    from lgd_amd import weights
    import models
    cfg = weights.CONFIGS[cfg_name]
    md = H.build_model_dict(cfg)
    md.tokenizer, md.text_encoder = FakeTokenizer(), FakeTextEncoder(cfg.cross_attention_dim, "cpu")
    models.model_dict = md
    models.models.model_dict = md
    return cfg, md


def record_phrase_calls(guidance, log):
    orig = guidance.get_phrase_indices

    def wrapped(tokenizer, prompt, phrases, *a, **k):
        out = orig(tokenizer, prompt, phrases, *a, **k)
        log.append(dict(prompt=prompt, phrases=list(phrases), words=list(k.get("words") or []),
                        add_suffix=bool(k.get("add_suffix_if_not_found", False)),
                        out=json.loads(json.dumps(out, default=lambda o: o.tolist() if hasattr(o, "tolist") else o))))
        return out
    guidance.get_phrase_indices = wrapped
    return orig


def run_lmd_plus():
    cfg, md = build("tiny_gligen")
    import generation.lmd_plus as g
    from utils import utils as ref_utils
    g.height = g.width = 256
    g.H = g.W = 32
    rec = dict(phrase_calls=[], gligen_calls=[], compose=[])
    g.sam.sam_refine_box = lambda sam_input_image, box, model_dict, verbose, H, W, **kw: (
        ref_utils.proportion_to_mask(box, H, W, return_np=True).astype(bool), 1.0)
    o_phr = record_phrase_calls(g.guidance, rec["phrase_calls"])
    o_gl, o_comp = g.pipelines.generate_gligen, g.latents.compose_latents_with_alignment

    # recorder around pipelines.latent_backward_guidance (looked up as a module global by generate_gligen at every
    # step, pipelines.py:418): its `latents` argument is the state at the START of step `index` — the teacher-forcing
    # points of tests/test_dropin_gpu.py::test_lmd_plus_overall_stage_teacher_forced
    o_bg = g.pipelines.latent_backward_guidance
    starts = []

    def bg(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k):
        starts.append((int(index), latents.detach().clone()))
        return o_bg(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k)
    g.pipelines.latent_backward_guidance = bg

    def gl(*a, **k):
        starts.clear()
        out = o_gl(*a, **k)
        rec["gligen_calls"].append(dict(latents_in=a[1].detach().clone(), out_latents=out[0].detach().clone(),
                                        latents_all=out[-1].detach().clone() if k.get("save_all_latents") else None,
                                        starts=[x for _, x in sorted(starts, key=lambda p: p[0])]))
        return out

    def comp(*a, **k):
        out = o_comp(*a, **k)
        rec["compose"].append((out[0].detach().clone(), out[1].detach().clone()))
        return out
    g.pipelines.generate_gligen, g.latents.compose_latents_with_alignment = gl, comp
    outs = {}
    for tag, spec, kw in (("a", SPEC, dict(bg_seed=3, fg_seed_start=3 + 123456789)),
                          ("b", SPEC3, dict(bg_seed=11, fg_seed_start=77, use_fast_schedule=True))):
        for v in rec.values():
            v.clear()
        r = g.run(spec, num_inference_steps=8, frozen_step_ratio=0.5, overall_max_index_step=3,
                  overall_max_iter=[2, 1, 1], overall_loss_threshold=0.0, use_autocast=False, **kw)
        n = len(spec["gen_boxes"])
        assert len(rec["gligen_calls"]) == n + 1 and len(rec["compose"]) == 1
        outs[f"{tag}_composed"] = rec["compose"][0][0].numpy()
        outs[f"{tag}_fg_idx"] = rec["compose"][0][1].numpy()
        outs[f"{tag}_final_latents"] = rec["gligen_calls"][-1]["out_latents"].numpy()
        ov = rec["gligen_calls"][-1]["starts"]
        assert len(ov) == 8, len(ov)                       # one call per step of the overall generation
        outs[f"{tag}_ov_starts"] = torch.stack(ov).numpy()
        for i in range(n):
            outs[f"{tag}_so{i}_latents_all"] = rec["gligen_calls"][i]["latents_all"].numpy()
        outs[f"{tag}_phrase_calls"] = np.array(json.dumps(rec["phrase_calls"]))
        outs[f"{tag}_image_shape"] = np.array(r.image.shape)
    g.pipelines.generate_gligen, g.latents.compose_latents_with_alignment = o_gl, o_comp
    g.pipelines.latent_backward_guidance = o_bg
    g.guidance.get_phrase_indices = o_phr
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "run_lmd_plus_tiny.npz"), **outs)
    print("wrote run_lmd_plus_tiny.npz", {k: getattr(v, "shape", None) for k, v in outs.items()})


def run_lmd():
    cfg, md = build("tiny")
    import generation.lmd as g
    from utils import utils as ref_utils
    g.height = g.width = 256
    g.H = g.W = 32
    rec = dict(phrase_calls=[], compose=[], final=[])
    g.sam.sam_refine_attn = lambda sam_input_image, token_attn_np, model_dict, height, width, H, W, **kw: (
        None, 1.0)
    # lmd.py refines with the attention map; the box is not passed to sam_refine_attn, so the stand-in is installed
    # one level up, where the box is known
    o_so = g.generate_single_object_with_box

    def so(prompt, box, *a, **k):
        g.sam.sam_refine_attn = lambda *aa, **kk: (ref_utils.proportion_to_mask(box, g.H, g.W, return_np=True).astype(bool), 1.0)
        return o_so(prompt, box, *a, **k)
    g.generate_single_object_with_box = so
    o_phr = record_phrase_calls(g.guidance, rec["phrase_calls"])
    o_comp, o_pf = g.latents.compose_latents_with_alignment, g.pipelines.generate_partial_frozen

    def comp(*a, **k):
        out = o_comp(*a, **k)
        rec["compose"].append((out[0].detach().clone(), out[1].detach().clone()))
        return out

    def pf(*a, **k):
        out = o_pf(*a, **k)
        rec["final"].append(out[0].detach().clone())
        return out
    g.latents.compose_latents_with_alignment, g.pipelines.generate_partial_frozen = comp, pf
    r = g.run(SPEC, bg_seed=3, fg_seed_start=99, num_inference_steps=12, max_index_step=2, max_iter=[1],
              loss_threshold=0.0, overall_max_index_step=3, overall_max_iter=[2, 1, 1], overall_loss_threshold=0.0,
              so_center_box=False, align_with_overall_bboxes=False, use_autocast=False)
    outs = dict(composed=rec["compose"][0][0].numpy(), fg_idx=rec["compose"][0][1].numpy(),
                final_latents=rec["final"][0].numpy(), phrase_calls=np.array(json.dumps(rec["phrase_calls"])),
                image_shape=np.array(r.image.shape))
    g.latents.compose_latents_with_alignment, g.pipelines.generate_partial_frozen = o_comp, o_pf
    g.guidance.get_phrase_indices = o_phr
    g.generate_single_object_with_box = o_so
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "run_lmd_tiny.npz"), **outs)
    print("wrote run_lmd_tiny.npz", {k: getattr(v, "shape", None) for k, v in outs.items()})


def run_backward_guidance():
    """The reference's own, unmodified `generation/backward_guidance.run` (backward_guidance.py:43-137): ONE guided
    generation whose energy is the RATIO-based branch of add_ca_loss_per_attn_map_to_loss (utils/guidance.py:118-130) —
    the plugin's kwargs (:99-112) carry no `use_ratio_based_loss`, so the function's default (True, :91) applies — and no
    reference-attention term (`ref_ca_saved_attns=None`).  Recorded: the latents entering every
    `latent_backward_guidance` call (teacher-forcing points), every value `compute_ca_lossv3` returned (one per guidance
    iteration, unscaled), the iteration count per step, the final latents."""
    cfg, md = build("tiny")
    import generation.backward_guidance as g
    g.height = g.width = 256
    g.H = g.W = 32
    g.num_inference_steps = 8
    outs = {}
    o_bg, o_loss, o_sg = g.pipelines.latent_backward_guidance, g.guidance.compute_ca_lossv3, g.pipelines.generate_semantic_guidance
    starts, losses, finals = [], [], []

    def bg(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k):
        starts.append(latents.detach().clone())
        n0 = len(losses)
        out = o_bg(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k)
        iters.append(len(losses) - n0)
        ends.append(out[0].detach().clone())
        return out

    def closs(*a, **k):
        assert "use_ratio_based_loss" not in k          # what this golden is about: the default branch runs
        out = o_loss(*a, **k)
        losses.append(float(out.detach()))
        return out

    def sg(*a, **k):
        call.update(latents_in=a[1].detach().clone(), text_embeddings=a[2][0].detach().clone(), bboxes=k["bboxes"],
                    object_positions=k["object_positions"])
        out = o_sg(*a, **k)
        finals.append(out[0].detach().clone())
        return out
    g.pipelines.latent_backward_guidance, g.guidance.compute_ca_lossv3, g.pipelines.generate_semantic_guidance = bg, closs, sg
    phrase_calls, call = [], {}
    o_phr = record_phrase_calls(g.guidance, phrase_calls)
    import warnings
    for tag, spec, kw in (("a", SPEC, dict(bg_seed=3, overall_loss_scale=30, overall_loss_threshold=0.2,
                                           overall_max_iter=5, overall_max_index_step=3)),
                          ("b", SPEC3, dict(bg_seed=11, overall_loss_scale=30, overall_loss_threshold=0.0,
                                            overall_max_iter=2, overall_max_index_step=2))):
        starts.clear(), losses.clear(), finals.clear(), phrase_calls.clear(), call.clear()
        iters, ends = [], []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = g.run(spec, **kw)
        assert len(starts) == 8 and len(finals) == 1
        outs[f"{tag}_starts"] = torch.stack(starts).numpy()
        outs[f"{tag}_guided"] = torch.stack(ends).numpy()          # latents leaving the guidance loop of each step
        outs[f"{tag}_iters"] = np.array(iters)
        outs[f"{tag}_losses"] = np.array(losses, dtype=np.float64)  # unscaled, in call order
        outs[f"{tag}_final_latents"] = finals[0].numpy()
        outs[f"{tag}_image_shape"] = np.array(r.image.shape)
        outs[f"{tag}_kwargs"] = np.array(json.dumps(kw))
        outs[f"{tag}_phrase_calls"] = np.array(json.dumps(phrase_calls))
        outs[f"{tag}_latents_in"] = call["latents_in"].numpy()
        outs[f"{tag}_text_embeddings"] = call["text_embeddings"].numpy()          # [uncond; cond] of the overall prompt
        outs[f"{tag}_bboxes"] = np.array(json.dumps(call["bboxes"]))
        outs[f"{tag}_object_positions"] = np.array(json.dumps(call["object_positions"]))
        print(tag, "iterations per step", iters, "losses", [round(x, 4) for x in losses])
    g.pipelines.latent_backward_guidance, g.guidance.compute_ca_lossv3, g.pipelines.generate_semantic_guidance = o_bg, o_loss, o_sg
    g.guidance.get_phrase_indices = o_phr
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "run_backward_guidance_tiny.npz"), **outs)
    print("wrote run_backward_guidance_tiny.npz", {k: getattr(v, "shape", None) for k, v in outs.items()})


def run_gligen():
    """The reference's own, unmodified `generation/gligen.run` (gligen.py:42-99): ONE generate_gligen call on the overall
    prompt with the per-box prompts as grounding phrases.  Recorded: the initial latents, the final latents."""
    cfg, md = build("tiny_gligen")
    import models
    models.sd_key = models.models.sd_key = "tiny_gligen"          # generation/gligen.py:16 asserts "gligen" in models.sd_key
    import generation.gligen as g
    g.height = g.width = 256
    g.H = g.W = 32
    g.num_inference_steps = 8
    rec = []
    o_gl = g.pipelines.generate_gligen

    def gl(*a, **k):
        out = o_gl(*a, **k)
        rec.append(dict(latents_in=a[1].detach().clone(), phrases=list(a[5]), bboxes=[list(b) for b in a[4]], out=out[0].detach().clone(),
                        beta=k.get("gligen_scheduled_sampling_beta"), guidance_scale=k.get("guidance_scale")))
        return out
    g.pipelines.generate_gligen = gl
    outs = {}
    for tag, spec, kw in (("a", SPEC, dict(bg_seed=3)), ("b", dict(SPEC3, gen_boxes=SPEC3["gen_boxes"][1:]), dict(bg_seed=11, gligen_scheduled_sampling_beta=0.25))):
        rec.clear()
        r = g.run(spec, **kw)
        assert len(rec) == 1
        outs[f"{tag}_latents_in"] = rec[0]["latents_in"].numpy()
        outs[f"{tag}_final_latents"] = rec[0]["out"].numpy()
        outs[f"{tag}_call"] = np.array(json.dumps(dict(phrases=rec[0]["phrases"], bboxes=rec[0]["bboxes"], beta=rec[0]["beta"],
                                                       guidance_scale=rec[0]["guidance_scale"])))
        outs[f"{tag}_image_shape"] = np.array(r.image.shape)
    g.pipelines.generate_gligen = o_gl
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "run_gligen_tiny.npz"), **outs)
    print("wrote run_gligen_tiny.npz", {k: getattr(v, "shape", None) for k, v in outs.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "lmd_plus"):
        run_lmd_plus()
    if which in ("all", "lmd"):
        run_lmd()
    if which in ("all", "backward_guidance"):
        run_backward_guidance()
    if which in ("all", "gligen"):
        run_gligen()
