"""CPU suite: host rewrites of the weight packing.  The LayerNorm fold (weightstore._lnlin, LGD_EPI_ROWNORM in
include/lgd_hip.h) must be an identity of the reference's LayerNorm -> Linear (attention.py:185,206,223): checked in
fp64 on the packed entries of a synthetic state dict, GEGLU row interleave included."""
import torch

import lgd_amd  # noqa: F401
from lgd_amd import weights
from lgd_amd.weightstore import WeightStore, geglu_perm


def test_layernorm_fold_entries_reproduce_layernorm_then_linear():
    cfg = weights.CONFIGS["tiny_gligen"]
    sd = weights.synth_state_dict(cfg, 3)
    ws = WeightStore(cfg, "cpu")
    ws.load_state_dict(sd)
    t = "mid_block.attentions.0.transformer_blocks.0"
    C = ws.h[f"{t}.attn2.to_q.w"].shape[1]
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(37, C, generator=g) * 1.3 + 2.0 * torch.randn(37, 1, generator=g)).double()
    mu, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    sites = [(f"{t}.attn1.qkv", f"{t}.norm1", False), (f"{t}.attn2.to_q", f"{t}.norm2", False),
             (f"{t}.ff.net.0.proj", f"{t}.norm3", True), (f"{t}.fuser.ff.net.0.proj", f"{t}.fuser.norm2", True)]
    for name, norm, geglu in sites:
        gm, bt = sd[f"{norm}.weight"].double(), sd[f"{norm}.bias"].double()
        assert (gm - 1).abs().max() > 1e-3 and bt.abs().max() > 1e-3, "synthetic LayerNorm parameters must be non-trivial"
        w = ws.h[f"{name}.w"].double()                                   # kernel row order, fp16 values
        b = ws.f[f"{name}.b"].double() if f"{name}.b" in ws.f else 0.0
        ref = ((x - mu) * rstd * gm + bt) @ w.t() + b
        wln, cs, bln = ws.h[f"{name}.wln"].double(), ws.f[f"{name}.cs"].double(), ws.f[f"{name}.bln"].double()
        got = rstd * (x @ wln.t() - mu * cs[None, :]) + bln[None, :]
        # fp16 rounding of W * gamma is the only difference (2^-11 relative per weight)
        assert (got - ref).abs().max() / ref.abs().max() < 2e-3, name
        assert torch.equal(ws.f[f"{name}.cs"], ws.h[f"{name}.wln"].float().sum(1)), "colsum must be the sums of the STORED fp16 weights"
        if geglu:
            perm = geglu_perm(w.shape[0] // 2)
            assert torch.equal(ws.h[f"{name}.w"], sd[f"{name}.weight"][perm].half())


def test_every_folded_site_has_its_three_entries():
    cfg = weights.CONFIGS["tiny_gligen"]
    ws = WeightStore(cfg, "cpu")
    names = {k[:-4] for k in ws.h if k.endswith(".wln")}
    assert names and all(f"{n}.cs" in ws.f and f"{n}.bln" in ws.f and f"{n}.w" in ws.h for n in names)
    assert all(ws.h[f"{n}.wln"].shape == ws.h[f"{n}.w"].shape for n in names)
    per_block = 4                                                         # qkv, to_q, ff, fuser ff
    n_blocks = sum(a.depth for b in ws.blocks for a in b.attns)
    assert len(names) == per_block * n_blocks


def test_store_without_the_fold_carries_no_twins():
    """LGD_FOLD_LN=0 engines (and anything else that never builds a folded no-grad plan) neither store nor broadcast the
    folded twins: same unfolded entries, smaller arenas, load_state_dict still fills everything it laid out."""
    cfg = weights.CONFIGS["tiny_gligen"]
    sd = weights.synth_state_dict(cfg, 3)
    full, lean = WeightStore(cfg, "cpu"), WeightStore(cfg, "cpu", fold_ln=False)
    lean.load_state_dict(sd)
    full.load_state_dict(sd)
    assert not any(k.endswith(".wln") for k in lean.h) and not any(k.endswith((".cs", ".bln")) for k in lean.f)
    assert lean.arena16.numel() < full.arena16.numel() and lean.arena32.numel() < full.arena32.numel()
    assert set(lean.h) == {k for k in full.h if not k.endswith(".wln")}
    for k in lean.h:
        assert torch.equal(lean.h[k], full.h[k]), k
