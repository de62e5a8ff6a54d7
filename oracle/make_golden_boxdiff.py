"""ORACLE TEST INFRASTRUCTURE (build container only) — BoxDiff goldens, from the reference's OWN code.

  1. tests/golden/boxdiff_energy.npz — utils/boxdiff.py:121-196 `compute_ca_loss_boxdiff` (unmodified, through
     oracle/ref_harness.py) on seeded probability maps of the five BoxDiff keys (tests/boxdiff_maps.py regenerates them):
     loss value and d loss / d map (one tensor: the same for every key and head); cases: 16x16 maps with two single-box phrases (the SD1.5 geometry), 8x8 maps (the tiny test network), a phrase
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

from boxdiff_maps import KEYS, energy_cases, make_maps  # noqa: E402  (tests/boxdiff_maps.py: shared with the tests)


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
        outs[f"{name}_spec"] = np.array(json.dumps(dict(side=c["side"], heads=c["heads"], bboxes=c["bboxes"], pos=c["pos"], seed=c["seed"])))
        # the energy sees the MEAN over keys and heads (utils/boxdiff.py:152): d loss / d map is one tensor, repeated for every
        # key and head — stored once (the maps themselves are regenerated from the seed: tests/boxdiff_maps.py)
        for i in range(1, len(KEYS)):
            assert torch.equal(grads[i], grads[0])
        assert all(torch.equal(grads[0][:, h], grads[0][:, 0]) for h in range(c["heads"]))
        outs[f"{name}_grad"] = grads[0][0, 0].numpy()                                  # [HW, 77]
        outs[f"{name}_map0_checksum"] = np.array(float(maps[KEYS[0]].double().sum()))
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
