"""ORACLE TEST INFRASTRUCTURE — UNet-forward golden for the SDXL-refiner BLOCK LAYOUT (DownBlock2D / UpBlock2D without
attention at the outermost AND innermost resolution, CrossAttn blocks in between; linear projections, 64-wide heads)
as far as the reference's own UNet2DConditionModel (diffusers-0.18 lineage) can build it: one transformer layer per
block, no text_time conditioning — those two are [ext]-only and stay "parity unpinned" (oracle/restate_sdxl.py).

    python oracle/make_golden_outer.py        # build container only; writes tests/golden/unet_fwd_tiny_outer.npz"""
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

L = 32
torch.set_num_threads(8)
cfg = weights.CONFIGS["tiny_outer"]
unet = rh.build_ref_unet(cfg, 0)
g = lambda shape, seed: torch.randn(shape, generator=torch.Generator().manual_seed(seed))
x = g((2, 4, L, L), 11)
unc, cond = weights.synth_embeddings(cfg, 1, seed=1)
ehs = torch.cat([unc, cond])
with torch.no_grad():
    eps = unet(x, torch.tensor(501), encoder_hidden_states=ehs).sample
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "unet_fwd_tiny_outer.npz"), x=x.numpy(), t=np.int64(501),
                    ehs=ehs.numpy(), eps=eps.numpy())
print("wrote unet_fwd_tiny_outer.npz; eps", tuple(eps.shape), float(eps.abs().max()))
