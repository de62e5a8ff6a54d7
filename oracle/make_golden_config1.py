"""ORACLE TEST INFRASTRUCTURE (build container only) — BASELINE config[0] run END TO END by the reference itself.

"SD1.5 fp32, 1 cached layout (2 boxes), 10 DDIM steps, training-free LMD, diffusers CPU scheduler — reference CPU path":
the reference's OWN, unmodified `generation/lmd.run` (generation/lmd.py:215-551, its default arguments: per-box AND
overall cross-attention guidance, max_index_step 30, reference-attention transfer, centred per-box boxes + alignment,
frozen_step_ratio 0.5, fp32 — use_autocast=False) on the FULL-WIDTH SD1.5 network (859 M parameters, seeded synthetic
weights of lgd_amd.weights: there are no checkpoints in the sandbox), CPU, through oracle/ref_harness.py, with
  * the fake whitespace tokenizer / table text encoder of tests/fake_text.py at width 768,
  * SAM replaced by the box mask,
  * recorder wrappers only (no behaviour change).
Writes tests/golden/run_lmd_sd15_config1.npz: for each of the three generations (two per-box, one overall) the latents
entering every denoising step (= the teacher-forcing points), the guidance iteration count and every guidance loss; the
composed latents, the foreground indices, the final latents; and the wall-clock of the whole run() on this container's
cores — the only measurement of the reference CPU pipeline on its own configuration anywhere in this repository
(profiles/r05_config1_reference_cpu.json).

    python oracle/make_golden_config1.py        # ~25-40 min on 8 cores
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
import ref_harness as H  # noqa: E402
from make_golden_runs import build, record_phrase_calls  # noqa: E402

# the canonical layout of configs 1 / 2 = demo entry 3 of the reference's cache, 512-px boxes (x, y, w, h)
SPEC = dict(prompt="A realistic image of a white deer and a gray bear in an empty factory scene",
            gen_boxes=[("a white deer", [74, 177, 183, 235]), ("a gray bear", [314, 193, 189, 216])],
            bg_prompt="A realistic image of an empty factory scene", extra_neg_prompt="")
STEPS = 10


def main():
    threads = int(os.environ.get("LGD_CONFIG1_THREADS", os.cpu_count() or 1))
    torch.set_num_threads(threads)
    cfg, md = build("sd15")
    import generation.lmd as g
    from utils import utils as ref_utils
    assert g.height == g.width == 512 and g.H == g.W == 64
    # As written the reference cannot run this configuration: get_token_attnv2 averages the maps saved from step
    # `attn_aggregation_step_start` = 10 on (generation/lmd.py:36,124-131; utils/attn.py:15-19), which is an empty stack
    # at 10 steps ("RuntimeError: stack expects a non-empty TensorList", reproduced here).  The module constant is
    # lowered to 5; its only consumer is the SAM point prompt, which the box-mask stand-in below ignores.
    g.attn_aggregation_step_start = 5
    rec = dict(phrase_calls=[], compose=[], gens=[])
    o_so = g.generate_single_object_with_box

    def so(prompt, box, *a, **k):        # lmd.py refines with the attention map; the box is known one level up
        g.sam.sam_refine_attn = lambda *aa, **kk: (ref_utils.proportion_to_mask(box, g.H, g.W, return_np=True).astype(bool), 1.0)
        return o_so(prompt, box, *a, **k)
    g.generate_single_object_with_box = so
    o_phr = record_phrase_calls(g.guidance, rec["phrase_calls"])
    o_comp, o_pf, o_sg = g.latents.compose_latents_with_alignment, g.pipelines.generate_partial_frozen, g.pipelines.generate_semantic_guidance
    o_bg, o_loss = g.pipelines.latent_backward_guidance, g.guidance.compute_ca_lossv3
    cur = dict(starts=[], iters=[], losses=[])

    def bg(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k):
        cur["starts"].append(latents.detach().clone())
        n0 = len(cur["losses"])
        out = o_bg(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k)
        cur["iters"].append(len(cur["losses"]) - n0)
        return out

    def closs(*a, **k):
        out = o_loss(*a, **k)
        cur["losses"].append(float(out.detach()))
        return out

    def wrap_gen(fn, kind):
        def inner(*a, **k):
            cur["starts"], cur["iters"], cur["losses"] = [], [], []
            t0 = time.time()
            out = fn(*a, **k)
            if kind == "overall":          # the reference maps of the transfer term, as lmd.run hands them over (aligned)
                rec["refs"] = k["semantic_guidance_kwargs"]["ref_ca_saved_attns"]
                rec["sg_kwargs"] = {kk: vv for kk, vv in k["semantic_guidance_kwargs"].items() if kk != "ref_ca_saved_attns"}
                rec["frozen"] = dict(frozen_steps=a[5], frozen_mask=a[2].detach().clone(), latents_all=a[1].detach().clone())
            else:
                rec.setdefault("so_kwargs", []).append(dict(k["semantic_guidance_kwargs"]))
                # round 6: the maps the per-box stage SAVES (pipelines.py:129-247, return_saved_cross_attn), before lmd.run
                # shifts them onto the overall boxes: per step a dict key -> [1, heads, HW, 1] (condition half, word token)
                sk = [tuple(kk_) for kk_ in k["saved_cross_attn_keys"]]
                rec.setdefault("so_saved", []).append(dict(keys=sk, maps={kk_: torch.stack([st[kk_][0, :, :, 0] for st in out[2]]).clone()
                                                                          for kk_ in dict.fromkeys(sk)}))
            rec["gens"].append(dict(kind=kind, seconds=time.time() - t0, starts=torch.stack(cur["starts"]), iters=list(cur["iters"]),
                                    losses=list(cur["losses"]), final=out[0].detach().clone(),
                                    latents_in=(a[1][0] if kind == "overall" else a[1]).detach().clone(),
                                    text_embeddings=(a[3][0] if kind == "overall" else a[2][0]).detach().clone(),
                                    # the per-box stage passes bboxes / phrases / object_positions positionally
                                    # (lmd.py:99-106), the overall stage by keyword (lmd.py:538-540)
                                    bboxes=k["bboxes"] if "bboxes" in k else a[4],
                                    object_positions=k["object_positions"] if "object_positions" in k else a[6]))
            return out
        return inner

    def comp(*a, **k):
        out = o_comp(*a, **k)
        rec["compose"].append((out[0].detach().clone(), out[1].detach().clone()))
        return out
    g.pipelines.latent_backward_guidance, g.guidance.compute_ca_lossv3 = bg, closs
    g.pipelines.generate_semantic_guidance = wrap_gen(o_sg, "so")
    g.pipelines.generate_partial_frozen = wrap_gen(o_pf, "overall")
    g.latents.compose_latents_with_alignment = comp
    kw = dict(bg_seed=0, fg_seed_start=123456789, num_inference_steps=STEPS)          # every other argument: lmd.py:215-256
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = g.run(SPEC, **kw)
    wall = time.time() - t0
    assert [x["kind"] for x in rec["gens"]] == ["so", "so", "overall"] and len(rec["compose"]) == 1
    outs = dict(kwargs=np.array(json.dumps(kw)), spec=np.array(json.dumps(SPEC)), wall_s=np.array(wall),
                threads=np.array(threads), composed=rec["compose"][0][0].numpy(), fg_idx=rec["compose"][0][1].numpy(),
                image_shape=np.array(r.image.shape), phrase_calls=np.array(json.dumps(rec["phrase_calls"])))
    for i, x in enumerate(rec["gens"]):
        outs[f"g{i}_starts"] = x["starts"].numpy()
        outs[f"g{i}_iters"] = np.array(x["iters"])
        outs[f"g{i}_losses"] = np.array(x["losses"], dtype=np.float64)           # unscaled compute_ca_lossv3 values
        outs[f"g{i}_final"] = x["final"].numpy()
        outs[f"g{i}_latents_in"] = x["latents_in"].numpy()
        outs[f"g{i}_text_embeddings"] = x["text_embeddings"].numpy()
        outs[f"g{i}_bboxes"] = np.array(json.dumps(x["bboxes"]))
        outs[f"g{i}_object_positions"] = np.array(json.dumps(x["object_positions"]))
        outs[f"g{i}_seconds"] = np.array(x["seconds"])
        print(f"generation {i} ({x['kind']}): {x['seconds']:.1f} s, guidance iterations per step {x['iters']}, "
              f"losses {[round(v, 4) for v in x['losses'][:6]]} ...")
    # overall stage: reference maps per object / step / key [T, heads, HW] (one box per object in this layout), the
    # frozen-blend inputs, the guidance kwargs of both stages
    keys = [tuple(kk) for kk in rec["sg_kwargs"]["guidance_attn_keys"]]
    for o, per_obj in enumerate(rec["refs"]):
        boxes = per_obj if isinstance(per_obj[0], list) else [per_obj]
        assert len(boxes) == 1
        for ki, kk in enumerate(keys):
            outs[f"ov_ref_o{o}_k{ki}"] = torch.stack([boxes[0][t][kk][0, :, :, 0] for t in range(STEPS)]).numpy()
    outs["ov_frozen_steps"] = np.array(rec["frozen"]["frozen_steps"])
    outs["ov_frozen_mask"] = rec["frozen"]["frozen_mask"].numpy()
    outs["ov_latents_all"] = rec["frozen"]["latents_all"].numpy()
    js = lambda d: json.dumps({kk: (vv if not torch.is_tensor(vv) else vv.tolist()) for kk, vv in d.items()}, default=lambda o_: list(o_))
    outs["ov_guidance_kwargs"] = np.array(js(rec["sg_kwargs"]))
    outs["so_guidance_kwargs"] = np.array(json.dumps([json.loads(js(d)) for d in rec["so_kwargs"]]))
    maps = {}
    for i, sv in enumerate(rec.get("so_saved", [])):
        maps[f"so{i}_saved_keys"] = np.array(json.dumps([list(kk_) for kk_ in sv["keys"]]))
        for ki, kk_ in enumerate(dict.fromkeys(sv["keys"])):
            maps[f"so{i}_saved_k{ki}"] = sv["maps"][kk_].numpy().astype(np.float16)      # [T, heads, HW]; fp16 halves the file
    if os.environ.get("LGD_CONFIG1_MAPS_ONLY"):
        # keep the committed golden (its gates were measured against it): this run must reproduce it bit for bit, then
        # only the per-box maps are written, beside it
        old = np.load(os.path.join(ROOT, "tests", "golden", "run_lmd_sd15_config1.npz"))
        same = all(np.array_equal(old[k_], outs[k_]) for k_ in ("g0_starts", "g1_starts", "g2_starts", "composed", "g2_final", "fg_idx"))
        print("re-run reproduces the committed golden bit for bit:", same)
        if same:
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", "run_lmd_sd15_config1_maps.npz"), **maps)
            return
    outs.update(maps)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "run_lmd_sd15_config1.npz"), **outs)
    summary = dict(what="BASELINE config[0]: the reference's own generation/lmd.run, SD1.5 architecture (seeded synthetic weights), "
                        "fp32, CPU, 1 cached layout (2 boxes), 10 DDIM steps, default arguments, SAM = box masks, VAE = stub",
                   host=f"{threads} torch threads of the build container ({os.cpu_count()} cores)", wall_s=round(wall, 1),
                   images_per_s=round(1.0 / wall, 6), generations=[dict(kind=x["kind"], seconds=round(x["seconds"], 1),
                                                                        unet_main_calls=STEPS, guidance_iterations=int(sum(x["iters"])))
                                                                   for x in rec["gens"]],
                   source="oracle/make_golden_config1.py")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(summary, open(os.path.join(ROOT, "profiles", "r05_config1_reference_cpu.json"), "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
