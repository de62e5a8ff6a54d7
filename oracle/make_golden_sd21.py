"""ORACLE TEST INFRASTRUCTURE — UNet-forward golden for an SD2.x-style configuration (linear proj_in/out,
per-level head counts giving 64-wide heads; BASELINE config 3 / SURVEY.md §8a U1,U4,A1) produced by the
reference's own UNet2DConditionModel through oracle/ref_harness.py, with the seeded synthetic weights.

    python oracle/make_golden_sd21.py        # build container only; writes tests/golden/unet_fwd_tiny_sd21.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import lgd_amd  # noqa: E402,F401
from lgd_amd import weights  # noqa: E402
import ref_harness as rh  # noqa: E402

KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]
OBJ_KEY = ("down", 2, 1, 0)
L = 32
torch.set_num_threads(8)
cfg = weights.CONFIGS["tiny_sd21"]
unet = rh.build_ref_unet(cfg, 0)
g = lambda shape, seed: torch.randn(shape, generator=torch.Generator().manual_seed(seed))
x = g((2, 4, L, L), 11)
unc, cond = weights.synth_embeddings(cfg, 1, seed=1)
ehs = torch.cat([unc, cond])
saved = {}
with torch.no_grad():
    eps = unet(x, torch.tensor(501), encoder_hidden_states=ehs,
               cross_attention_kwargs={"save_attn_to_dict": saved, "save_keys": [OBJ_KEY, *KEYS]}).sample
arrs = dict(x=x.numpy(), t=np.int64(501), ehs=ehs.numpy(), eps=eps.numpy())
for k, v in saved.items():
    arrs["map_" + "_".join(str(i) for i in k)] = v.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "unet_fwd_tiny_sd21.npz"), **arrs)
print("wrote unet_fwd_tiny_sd21.npz; eps", tuple(eps.shape), {k: tuple(v.shape) for k, v in saved.items()})
