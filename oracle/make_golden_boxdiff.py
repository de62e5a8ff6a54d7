"""ORACLE TEST INFRASTRUCTURE (build container only) — BoxDiff goldens, from the reference's OWN code.

  1. tests/golden/boxdiff_energy.npz — utils/boxdiff.py:121-196 `compute_ca_loss_boxdiff` (unmodified, through
     oracle/ref_harness.py) on seeded probability maps of the five BoxDiff keys: loss value and d loss / d map for every
     key; cases: 16x16 maps with two single-box phrases (the SD1.5 geometry), 8x8 maps (the tiny test network), a phrase
     with two boxes, a box covering most of the map, and a box so small that its inner-box top-k has k = 0.
  2. tests/golden/run_boxdiff_tiny.npz — the reference's own `generation/boxdiff.run` (generation/boxdiff.py:46-131) on
     the tiny network with the fake tokenizer / text encoder of tests/fake_text.py: the latents entering every
     `latent_backward_guidance_boxdiff` call, the latents leaving it, its loss, the final latents, the inputs.

    python oracle/make_golden_boxdiff.py
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness as H  # noqa: E402
from make_golden_runs import SPEC, SPEC3, build, record_phrase_calls  # noqa: E402

KEYS = [("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]    # generation/boxdiff.py:33-39


def energy_cases():
    return dict(
        hw256=dict(side=16, heads=8, bboxes=[[0.15, 0.35, 0.5, 0.8], [0.6, 0.38, 0.98, 0.8]], pos=[[1, 2, 3], [5, 6, 7]], seed=0),
        hw64=dict(side=8, heads=4, bboxes=[[0.1, 0.3, 0.45, 0.85], [0.55, 0.3, 0.95, 0.8]], pos=[[1, 2, 3], [5, 6, 7]], seed=1),
        two_boxes=dict(side=16, heads=8, bboxes=[[[0.05, 0.5, 0.3, 0.9], [0.4, 0.45, 0.7, 0.85]], [[0.72, 0.1, 0.97, 0.4]]],
                       pos=[[2, 3], [9]], seed=2),
        edge=dict(side=16, heads=8, bboxes=[[0.0, 0.0, 1.0, 0.6], [0.3, 0.7, 0.62, 0.97]], pos=[[4], [6, 7]], seed=3),
        # a 2 x 2-pixel box: (mask.sum() * P).long() = 0 -> top-k of ZERO elements, mean = NaN, and Python's
        # max(0, 1 - nan) = 0 drops the inner-box term (utils/boxdiff.py:81-83,107)
        tiny_box=dict(side=16, heads=8, bboxes=[[0.5, 0.5, 0.62, 0.62], [0.1, 0.2, 0.4, 0.9]], pos=[[2, 3], [8]], seed=4),
    )


def make_maps(side, heads, seed):
    """Five maps [1, heads, HW, 77] of probabilities over the 77 text tokens with a spatial structure (so that the
    token soft-max at x100 is not one-hot everywhere and the top-k selections are not degenerate)."""
    g = torch.Generator().manual_seed(seed)
    hw = side * side
    out = {}
    yy, xx = torch.meshgrid(torch.linspace(0, 1, side), torch.linspace(0, 1, side), indexing="ij")
    for k in KEYS:
        logits = torch.randn((1, heads, hw, 77), generator=g) * 0.3
        for tok in range(1, 12):                                        # smooth bumps per token, different per head
            cx, cy = torch.rand(heads, generator=g), torch.rand(heads, generator=g)
            bump = torch.exp(-(((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) / 0.05))
            logits[0, :, :, tok] += 2.0 * bump.reshape(heads, hw)
        out[k] = logits.softmax(dim=-1)
    return out


def golden_energy():
    H.setup()
    from utils import boxdiff
    outs = {}
    for name, c in energy_cases().items():
        maps = make_maps(c["side"], c["heads"], c["seed"])
        leaves = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss = boxdiff.compute_ca_loss_boxdiff(saved_attn=leaves, bboxes=c["bboxes"], object_positions=c["pos"],
                                                   guidance_attn_keys=KEYS, ref_ca_saved_attns=None, index=0, verbose=False)
        grads = torch.autograd.grad(loss, [leaves[k] for k in KEYS])
        outs[f"{name}_loss"] = np.array(float(loss))
        outs[f"{name}_spec"] = np.array(json.dumps(dict(side=c["side"], heads=c["heads"], bboxes=c["bboxes"], pos=c["pos"])))
        for i, k in enumerate(KEYS):
            outs[f"{name}_map{i}"] = maps[k].numpy()
            outs[f"{name}_grad{i}"] = grads[i].numpy()
        print(name, "loss", float(loss), "grad norms", [round(float(gm.norm()), 5) for gm in grads])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "boxdiff_energy.npz"), **outs)
    print("wrote boxdiff_energy.npz")


def golden_run():
    cfg, md = build("tiny")
    import generation.boxdiff as g
    g.height = g.width = 256
    g.H = g.W = 32
    g.num_inference_steps = 8
    g.verbose = False
    outs = {}
    o_bd, o_sg = g.pipelines.boxdiff.latent_backward_guidance_boxdiff, g.pipelines.generate_semantic_guidance
    starts, ends, losses, finals, call = [], [], [], [], {}

    def bd(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k):
        starts.append(latents.detach().clone())
        out = o_bd(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss, **k)
        ends.append(out[0].detach().clone())
        losses.append(float(out[1].detach()))
        call["kwargs"] = {kk: vv for kk, vv in k.items() if kk not in ("cross_attention_kwargs",)}
        return out

    def sg(*a, **k):
        assert k.get("use_boxdiff") is True
        call.update(latents_in=a[1].detach().clone(), text_embeddings=a[2][0].detach().clone(), bboxes=k["bboxes"],
                    object_positions=k["object_positions"])
        out = o_sg(*a, **k)
        finals.append(out[0].detach().clone())
        return out
    g.pipelines.boxdiff.latent_backward_guidance_boxdiff, g.pipelines.generate_semantic_guidance = bd, sg
    phrase_calls = []
    o_phr = record_phrase_calls(g.guidance, phrase_calls)
    for tag, spec, kw in (("a", SPEC, dict(bg_seed=3, overall_max_index_step=5)),
                          ("b", SPEC3, dict(bg_seed=11, overall_max_index_step=3))):
        starts.clear(), ends.clear(), losses.clear(), finals.clear(), phrase_calls.clear(), call.clear()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = g.run(spec, **kw)
        assert len(starts) == 8 and len(finals) == 1
        outs[f"{tag}_starts"] = torch.stack(starts).numpy()
        outs[f"{tag}_guided"] = torch.stack(ends).numpy()
        outs[f"{tag}_losses"] = np.array(losses, dtype=np.float64)      # amp-scaled (x10), as the function returns it
        outs[f"{tag}_final_latents"] = finals[0].numpy()
        outs[f"{tag}_image_shape"] = np.array(r.image.shape)
        outs[f"{tag}_kwargs"] = np.array(json.dumps(kw))
        outs[f"{tag}_phrase_calls"] = np.array(json.dumps(phrase_calls))
        outs[f"{tag}_latents_in"] = call["latents_in"].numpy()
        outs[f"{tag}_text_embeddings"] = call["text_embeddings"].numpy()
        outs[f"{tag}_bboxes"] = np.array(json.dumps(call["bboxes"]))
        outs[f"{tag}_object_positions"] = np.array(json.dumps(call["object_positions"]))
        outs[f"{tag}_guidance_kwargs"] = np.array(json.dumps({k: (v if not torch.is_tensor(v) else v.tolist())
                                                              for k, v in call["kwargs"].items() if k != "ref_ca_saved_attns"},
                                                             default=lambda o: list(o)))
        print(tag, "losses", [round(x, 4) for x in losses])
    g.pipelines.boxdiff.latent_backward_guidance_boxdiff, g.pipelines.generate_semantic_guidance = o_bd, o_sg
    g.guidance.get_phrase_indices = o_phr
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "run_boxdiff_tiny.npz"), **outs)
    print("wrote run_boxdiff_tiny.npz", {k: getattr(v, "shape", None) for k, v in outs.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "energy"):
        golden_energy()
    if which in ("all", "run"):
        golden_run()
