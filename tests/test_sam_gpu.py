"""SAM on the HIP kernels (lgd_amd/sam.py, csrc/sam.hip) vs (a) the torch statement of the new
kernels (tests/ops_emul.py) and (b) the Hugging Face `SamModel` the reference calls (models/sam.py:39-40), with seeded
random parameters (no checkpoints in the sandbox), fp32.  Tolerances: fp16 storage / fp32 accumulation vs fp32,
relative to the tensor max; mask logits additionally by sign agreement (what `post_process_masks` thresholds)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import lgd_amd  # noqa: E402,F401
import ops_emul  # noqa: E402
import sam_cases  # noqa: E402
from lgd_amd import ops  # noqa: E402
from lgd_amd import sam as lsam  # noqa: E402


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("B,Hs,window,NH,d,DA", [(2, 8, 3, 2, 32, 64), (1, 64, 14, 12, 64, 96), (1, 16, 0, 4, 32, 64),
                                                  (1, 64, 0, 2, 64, 192)])
def test_relpos_qkv_and_window_merge_kernels(dev, B, Hs, window, NH, d, DA):
    g = torch.Generator().manual_seed(0)
    S = window or Hs
    C = NH * d
    qkv = torch.randn(B * Hs * Hs, 3 * C, generator=g).half()
    bias = torch.randn(3 * C, generator=g)
    rel_h, rel_w = torch.randn(2 * S - 1, d, generator=g) * 0.3, torch.randn(2 * S - 1, d, generator=g) * 0.3
    scale = d ** -0.5
    want = ops_emul.sam_relpos_qkv(qkv, bias, rel_h, rel_w, B, Hs, Hs, window, NH, d, DA, scale)
    got = ops.sam_relpos_qkv(qkv.to(dev), bias.to(dev), rel_h.to(dev), rel_w.to(dev), B, Hs, Hs, window, NH, d, DA, scale)
    nwin = -(-Hs // S)
    valid = torch.zeros(B, nwin, nwin, S, S, dtype=torch.bool)
    yy = (torch.arange(nwin)[:, None] * S + torch.arange(S)[None, :])
    valid[:] = ((yy < Hs)[:, None, :, None] & (yy < Hs)[None, :, None, :])[None]
    valid = valid.reshape(-1)
    # K' and V' (one-hot columns, padding = projection bias) are exact copies; Q' bias columns are fp32 dot products
    # rounded once to fp16.  Q' of padded positions is never read back (window_merge drops those rows).
    assert torch.equal(got[1].cpu(), want[1]) and torch.equal(got[2].cpu(), want[2])
    assert relerr(got[0].cpu()[valid], want[0][valid]) < 2e-3
    oa = torch.randn(want[0].shape, generator=g).half()
    assert torch.equal(ops.sam_window_merge(oa.to(dev), B, Hs, Hs, window, NH, d, DA).cpu(),
                       ops_emul.sam_window_merge(oa, B, Hs, Hs, window, NH, d, DA))


def test_activation_kernels(dev):
    x = (torch.randn(4096 + 8, generator=torch.Generator().manual_seed(0)) * 3).half()
    assert relerr(ops.act(x.to(dev), ops.ACT_GELU), torch.nn.functional.gelu(x.float())) < 1e-3
    assert torch.equal(ops.act(x.to(dev), ops.ACT_RELU).cpu(), torch.relu(x))


@pytest.mark.parametrize("B,H,Sq,Sk,d", [(2, 12, 4096, 4096, 192), (25, 12, 196, 196, 96), (3, 8, 7, 4096, 16),
                                          (3, 8, 4096, 7, 16), (3, 8, 7, 7, 32)])
def test_attention_shapes_of_sam(dev, B, H, Sq, Sk, d):
    """Flash attention at the head widths / sequence lengths SAM uses (192 = 64 + 2 x 64 bias columns is new)."""
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B * Sq, H * d, generator=g).half().to(dev)
    k = torch.randn(B * Sk, H * d, generator=g).half().to(dev)
    v = torch.randn(B * Sk, H * d, generator=g).half().to(dev)
    o = torch.empty_like(q)
    scale = (64 if d > 64 else d) ** -0.5
    ops.attn_fwd(q, k, v, o, B, H, Sq, Sk, d, scale)
    sp = lambda t, S: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    ref = torch.nn.functional.scaled_dot_product_attention(sp(q, Sq), sp(k, Sk), sp(v, Sk), scale=scale)
    assert relerr(o, ref.permute(0, 2, 1, 3).reshape(B * Sq, H * d)) < 4e-3


def test_attention_with_very_negative_logits(dev):
    """Every logit far below -128 in the log2 domain: the first-tile reference shift of the narrow-head kernel must not
    rescale its (still zero) accumulators by exp2(+large) = inf (found with a random-weight SAM decoder)."""
    B, H, Sq, Sk, d = 2, 8, 256, 7, 16
    g = torch.Generator().manual_seed(0)
    q = (8 + torch.rand(B * Sq, H * d, generator=g)).half().to(dev)
    k = -(8 + torch.rand(B * Sk, H * d, generator=g)).half().to(dev)
    v = torch.randn(B * Sk, H * d, generator=g).half().to(dev)
    o = torch.empty_like(q)
    ops.attn_fwd(q, k, v, o, B, H, Sq, Sk, d, 0.25)
    sp = lambda t, S: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    ref = torch.nn.functional.scaled_dot_product_attention(sp(q, Sq), sp(k, Sk), sp(v, Sk), scale=0.25)
    assert torch.isfinite(o).all()
    assert relerr(o, ref.permute(0, 2, 1, 3).reshape(B * Sq, H * d)) < 2e-2


def _compare(dev, cfg, B, P, points, tol_emb, tol_mask, tol_iou, agree_min):
    transformers = pytest.importorskip("transformers")
    hf = sam_cases.build_hf(transformers, cfg)
    inp = sam_cases.inputs(cfg, B=B, P=P, points=points)
    mine = lsam.HipSamModel(lsam.SamConfig.from_hf(cfg), hf.state_dict(), device=dev)
    hf = hf.to(dev)
    with torch.no_grad():
        want = hf(**{k: v.to(dev) for k, v in inp.items()})
        want_emb = hf.get_image_embeddings(inp["pixel_values"].to(dev))
    got = mine(**inp)
    torch.cuda.synchronize()
    e_emb = relerr(mine.get_image_embeddings(inp["pixel_values"]), want_emb)
    e_mask, e_iou = relerr(got.pred_masks, want.pred_masks), relerr(got.iou_scores, want.iou_scores)
    agree = float(((got.pred_masks > 0) == (want.pred_masks > 0)).float().mean())
    print(f"[sam] emb {e_emb:.2e} masks {e_mask:.2e} iou {e_iou:.2e} sign agreement {agree:.4f}")
    assert got.pred_masks.shape == want.pred_masks.shape and got.pred_masks.dtype == torch.float32
    assert e_emb < tol_emb and e_mask < tol_mask and e_iou < tol_iou and agree > agree_min


@pytest.mark.parametrize("points", [False, True])
def test_sam_small_vs_transformers(dev, points):
    transformers = pytest.importorskip("transformers")
    _compare(dev, sam_cases.small_config(transformers), 2, 2, points, 5e-3, 2e-2, 1e-2, 0.995)   # measured 1e-3 / 3e-3 / 1.3e-3 / 0.9985


def test_sam_vit_base_vs_transformers(dev):
    """facebook/sam-vit-base geometry (1024^2 image, 64x64 tokens, 14x14 windows, global blocks 2/5/8/11), two boxes."""
    transformers = pytest.importorskip("transformers")
    _compare(dev, transformers.SamConfig(), 1, 2, False, 5e-3, 3e-2, 3e-2, 0.995)           # measured 1.4e-3 / 6.4e-3 / 6.2e-3 / 0.9987


def test_refinement_replay_on_hip_vs_reference_golden(dev, monkeypatch):
    """The calls oracle/make_golden_sam.py made on the reference's own models/sam.py (sam_refine_box, the batched
    sam_refine_boxes, sam_refine_attn with the point prompt) replayed through the drop-in module with the HIP model.
    fp16 kernels vs the fp32 CPU run: a handful of the 4096 latent-grid pixels sit within rounding of the threshold."""
    import os
    import sam_refine_checks
    transformers = pytest.importorskip("transformers")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(os.path.join(root, "llm-groundeddiffusion_amd", "dropin"))
    from models import sam as dsam
    md = dsam.wrap_sam(sam_cases.build_refine_hf(transformers), device=dev)
    worst = sam_refine_checks.replay(dsam, md, min_agree=0.95, conf_tol=3e-2)       # measured worst 0.973
    print(f"[sam refine] worst mask agreement with the reference's selection {worst:.4f}")
