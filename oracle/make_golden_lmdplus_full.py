"""ORACLE TEST INFRASTRUCTURE (build container only) — the HEADLINE configuration run END TO END by the reference itself.

BASELINE config[1]'s method (LMD+ = attention guidance + GLIGEN) on the network the metric is quoted on: the
reference's OWN, unmodified `generation/lmd_plus.run` (generation/lmd_plus.py:193-520) with every default argument
(per-box generations unguided with GLIGEN, overall generation guided: `overall_max_index_step` 30, `overall_max_iter`
4/3/2/2/1, loss threshold 5.0, reference-attention transfer on, `frozen_step_ratio` 0.5, scheduled-sampling beta 0.4)
on the FULL-WIDTH `sd14_gligen` network (SD1.4 UNet + GLIGEN fuser layers + position net, seeded synthetic weights of
lgd_amd.weights — there are no checkpoints in the sandbox), fp32 on CPU (`use_autocast`'s CUDA autocast has nothing to
act on here), 512 x 512, 20 DDIM steps, through oracle/ref_harness.py with
  * the fake whitespace tokenizer / table text encoder of tests/fake_text.py at width 768,
  * SAM replaced by the box mask,
  * recorder wrappers only (no behaviour change).
Writes tests/golden/run_lmd_plus_sd14gligen_full.npz: the two per-box histories (`latents_all`), the composed latents and
foreground indices, and for the overall generation the latents entering every denoising step (the teacher-forcing
points), the guidance iteration count of every step, every guidance loss, the final latents; plus the wall-clock of
the run on this container's cores (profiles/r06_config2_reference_cpu.json).

    python oracle/make_golden_lmdplus_full.py        # ~20-40 min on 8 cores
"""
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness as H  # noqa: E402,F401
from make_golden_runs import build, record_phrase_calls  # noqa: E402

# the canonical layout of configs 1 / 2 = demo entry 3 of the reference's cache, 512-px boxes (x, y, w, h)
SPEC = dict(prompt="A realistic image of a white deer and a gray bear in an empty factory scene",
            gen_boxes=[("a white deer", [74, 177, 183, 235]), ("a gray bear", [314, 193, 189, 216])],
            bg_prompt="A realistic image of an empty factory scene", extra_neg_prompt="")
STEPS = int(os.environ.get("LGD_FULL_STEPS", "20"))


def main():
    threads = int(os.environ.get("LGD_FULL_THREADS", os.cpu_count() or 1))
    torch.set_num_threads(threads)
    cfg, md = build("sd14_gligen")
    import generation.lmd_plus as g
    from utils import utils as ref_utils
    assert g.height == g.width == 512 and g.H == g.W == 64
    rec = dict(phrase_calls=[], gligen_calls=[], compose=[])
    g.sam.sam_refine_box = lambda sam_input_image, box, model_dict, verbose, H, W, **kw: (
        ref_utils.proportion_to_mask(box, H, W, return_np=True).astype(bool), 1.0)
    record_phrase_calls(g.guidance, rec["phrase_calls"])
    o_gl, o_comp = g.pipelines.generate_gligen, g.latents.compose_latents_with_alignment
    o_bg, o_loss = g.pipelines.latent_backward_guidance, g.guidance.compute_ca_lossv3
    cur = dict(starts=[], iters=[], losses=[])

    # latent_backward_guidance is looked up as a module global by generate_gligen at every step whose index is below
    # max_index_step (pipelines.py:418): its `latents` argument is the state at the START of step `index`
    def bg(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k):
        cur["starts"].append((int(index), latents.detach().clone()))
        n0 = len(cur["losses"])
        out = o_bg(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k)
        cur["iters"].append((int(index), len(cur["losses"]) - n0))
        return out

    def closs(*a, **k):
        out = o_loss(*a, **k)
        cur["losses"].append(float(out.detach()))
        return out

    def gl(*a, **k):
        cur["starts"], cur["iters"], cur["losses"] = [], [], []
        t0 = time.time()
        out = o_gl(*a, **k)
        sg = k.get("semantic_guidance_kwargs") or {}
        rec["gligen_calls"].append(dict(
            seconds=time.time() - t0, latents_in=a[1].detach().clone(), out_latents=out[0].detach().clone(),
            latents_all=out[-1].detach().clone() if k.get("save_all_latents") else None,
            starts=[x for _, x in sorted(cur["starts"], key=lambda p: p[0])], iters=[n for _, n in sorted(cur["iters"])],
            losses=list(cur["losses"]),
            sg_kwargs={kk: vv for kk, vv in sg.items() if kk != "ref_ca_saved_attns"},
            has_ref=sg.get("ref_ca_saved_attns") is not None,
            scalars={kk: k[kk] for kk in ("num_inference_steps", "gligen_scheduled_sampling_beta", "frozen_steps", "guidance_scale")
                     if kk in k}))
        return out

    def comp(*a, **k):
        out = o_comp(*a, **k)
        rec["compose"].append((out[0].detach().clone(), out[1].detach().clone()))
        return out
    g.pipelines.latent_backward_guidance, g.guidance.compute_ca_lossv3 = bg, closs
    g.pipelines.generate_gligen, g.latents.compose_latents_with_alignment = gl, comp
    kw = dict(bg_seed=0, fg_seed_start=123456789, num_inference_steps=STEPS, use_autocast=False)   # every other argument: lmd_plus.py:193-233
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = g.run(SPEC, **kw)
    wall = time.time() - t0
    n = len(SPEC["gen_boxes"])
    assert len(rec["gligen_calls"]) == n + 1 and len(rec["compose"]) == 1
    ov = rec["gligen_calls"][-1]
    outs = dict(kwargs=np.array(json.dumps(kw)), spec=np.array(json.dumps(SPEC)), wall_s=np.array(wall), threads=np.array(threads),
                composed=rec["compose"][0][0].numpy(), fg_idx=rec["compose"][0][1].numpy(),
                final_latents=ov["out_latents"].numpy(), ov_starts=torch.stack(ov["starts"]).numpy(),
                ov_iters=np.array(ov["iters"]), ov_losses=np.array(ov["losses"], dtype=np.float64),
                ov_latents_in=ov["latents_in"].numpy(), image_shape=np.array(r.image.shape),
                phrase_calls=np.array(json.dumps(rec["phrase_calls"])),
                ov_guidance_kwargs=np.array(json.dumps(ov["sg_kwargs"], default=lambda o_: o_.tolist() if hasattr(o_, "tolist") else list(o_))),
                ov_has_ref=np.array(ov["has_ref"]),
                ov_scalars=np.array(json.dumps(ov["scalars"], default=float)))
    for i in range(n):
        outs[f"so{i}_latents_all"] = rec["gligen_calls"][i]["latents_all"].numpy()
        outs[f"so{i}_seconds"] = np.array(rec["gligen_calls"][i]["seconds"])
        assert sum(rec["gligen_calls"][i]["iters"]) == 0          # per-box generations are unguided by default (max_index_step 0)
    print(f"overall generation: {ov['seconds']:.1f} s, guidance iterations per step {ov['iters']}, losses {[round(v, 4) for v in ov['losses'][:8]]} ...")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "run_lmd_plus_sd14gligen_full.npz"), **outs)
    summary = dict(what="BASELINE config[1]'s method on the reference CPU path: the reference's own generation/lmd_plus.run, sd14_gligen "
                        f"architecture (seeded synthetic weights), fp32, CPU, 1 cached layout (2 boxes), {STEPS} DDIM steps, default "
                        "arguments, SAM = box masks, VAE = stub",
                   host=f"{threads} torch threads of the build container ({os.cpu_count()} cores)", wall_s=round(wall, 1),
                   images_per_s=round(1.0 / wall, 6),
                   generations=[dict(kind="per-box" if i < n else "overall", seconds=round(c["seconds"], 1), unet_main_calls=STEPS,
                                     guidance_iterations=int(sum(c["iters"]))) for i, c in enumerate(rec["gligen_calls"])],
                   source="oracle/make_golden_lmdplus_full.py")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(summary, open(os.path.join(ROOT, "profiles", "r06_config2_reference_cpu.json"), "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
