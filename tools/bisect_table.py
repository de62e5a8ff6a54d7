"""GPU-box tool: which tuning-table entries move the full-width guidance gradient away from the oracle?  Runs the body
of tests/test_bench_path_gpu.py::test_fullsize_guidance_iteration_vs_oracle under variants of the latency table: the
committed one, an older one (argv[1]), and the committed one with the changed entries of one NEW tile code at a time
reverted to the older choice."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import lgd_amd  # noqa
from lgd_amd import ops
from lgd_amd.sampler import LMDSampler
from lgd_amd.scheduler import DDIMScheduler
import test_bench_path_gpu as T
import restate as R

dev = torch.device("cuda:0")
f = T.full(dev)
cfg, eng = f["cfg"], f["eng"]
x, _, cond, gl = T._inputs(cfg, dev)
guid = dict(bboxes=T.BOXES, object_positions=T.OBJ_POS, loss_scale=5, loss_threshold=0.0, max_iter=1, max_index_step=30,
            guidance_attn_keys=T.KEYS, use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)
rs = R.DDIM(); rs.set_timesteps(50)
tr_ref = []
R.latent_backward_guidance(f["sd"], f["cd"], rs, cond, 1, T.BOXES, T.OBJ_POS, rs.timesteps[1], x[:1].clone(), torch.tensor(1e4),
                           loss_scale=5, loss_threshold=0.0, max_iter=1, max_index_step=30, guidance_attn_keys=T.KEYS,
                           use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0,
                           gligen=dict(boxes=gl[0][:1].cpu(), positive_embeddings=gl[1][:1].cpu(), masks=gl[2][:1].cpu()), trace=tr_ref)
b = tr_ref[0]["grad"].double().reshape(-1)


def run(table, label):
    ops._TUNING["latency"] = table
    eng._plans.clear()
    sm = LMDSampler(eng, DDIMScheduler())
    tr = []
    with ops.desc_log() as log:
        sm.guidance_only(x[:1], cond, 50, 1, guid, gligen=gl, fuser=True, trace=tr)
    a = tr[0]["grad"].cpu().double().reshape(-1)
    cos = float(a @ b / (a.norm() * b.norm()))
    print(f"{label:60s} cosine {cos:.5f} rel-L2 {float((a - b).norm() / b.norm()):.3e} loss {tr[0]['loss']:.4f}", flush=True)
    return {k for k, _, _ in log}


new = json.load(open(os.path.join(ROOT, "llm-groundeddiffusion_amd", "tuning_gfx950.json")))
old = json.load(open(sys.argv[1]))
used = run(dict(new), "committed table")
run(dict(old), "older table")
ch = [k for k in new if k in old and k in used and (old[k]["tile"], old[k]["splits"]) != (new[k]["tile"], new[k]["splits"])]
print(len(ch), "changed entries are used by this plan")
for k in ch:
    t = dict(new); t[k] = old[k]
    run(t, f"revert {k} {new[k]['tile']}/{new[k]['splits']} -> {old[k]['tile']}/{old[k]['splits']}")
