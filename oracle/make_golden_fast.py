"""ORACLE TEST INFRASTRUCTURE — end-to-end golden of the optional fast schedule: the reference's own
`pipelines.generate_gligen(..., dynamic_num_inference_steps=True, fast_after_steps=4, fast_rate=2)`
(pipelines.py:323-473 with :358-359, :439-440, :449; the per-box call of lmd_plus.py:96-111) on the tiny
GLIGEN configuration with the seeded synthetic weights, through oracle/ref_harness.py.

    python oracle/make_golden_fast.py         # build container only; writes tests/golden/fast_tiny_gligen.npz"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import lgd_amd  # noqa: E402,F401
from lgd_amd import weights  # noqa: E402
import ref_harness as rh  # noqa: E402
from make_golden import BBOXES, KEYS, OBJ_KEY, L, FakeTokenizer, FakeTextEncoder, seeded  # noqa: E402

torch.set_num_threads(8)
mods = rh.ref_modules()
pipelines = mods["pipelines"]
cfg = weights.CONFIGS["tiny_gligen"]
md = rh.build_model_dict(cfg, 0)
pe = seeded((2, 768), 3)
md.tokenizer, md.text_encoder = FakeTokenizer(), FakeTextEncoder(pe)
unc, cond = weights.synth_embeddings(cfg, 1, seed=1)
ehs = torch.cat([unc, cond])
lat0 = seeded((1, 4, L, L), 41)
T, FAST_AFTER = 10, 4
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ret = pipelines.generate_gligen(
        md, lat0, (ehs, unc, cond), T, BBOXES, ["a", "b"], gligen_scheduled_sampling_beta=0.5,
        return_saved_cross_attn=True, saved_cross_attn_keys=[OBJ_KEY, *KEYS], return_cond_ca_only=True,
        return_token_ca_only=7, save_all_latents=True, show_progress=False,
        dynamic_num_inference_steps=True, fast_after_steps=FAST_AFTER, fast_rate=2)
lat, _, saved, lat_all = ret
print("steps run:", len(saved), "history rows:", len(lat_all), "scheduler timesteps:", md.scheduler.timesteps.tolist())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fast_tiny_gligen.npz"), ehs=ehs.numpy(), lat0=lat0.numpy(),
                    phrase_emb=pe.numpy(), T=np.int64(T), fast_after=np.int64(FAST_AFTER),
                    timesteps=md.scheduler.timesteps.numpy(), latents=lat.detach().numpy(),
                    latents_all=torch.stack(list(lat_all)).numpy() if not torch.is_tensor(lat_all) else lat_all.numpy(),
                    n_saved=np.int64(len(saved)), saved_up11_last=saved[-1][("up", 1, 1, 0)].numpy())
print("wrote tests/golden/fast_tiny_gligen.npz")
