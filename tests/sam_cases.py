"""Shared by the SAM tests: Hugging Face `SamModel`s with seeded random parameters (no checkpoints in the sandbox).
HF initialises the position tables (pos_embed, rel_pos_h / rel_pos_w) to ZERO and most weights at std 1e-10 scale for
the vision tower (`initializer_range`), which would leave the relative-position path untested and every activation
degenerate — so every parameter is re-drawn here at a scale that keeps activations O(1) through the network."""
import torch


def small_config(transformers):
    """8x8 token grid, 3x3 windows (padded to 9x9: exercises SAM's post-LayerNorm zero padding), one global block."""
    v = transformers.SamVisionConfig(hidden_size=64, output_channels=64, num_hidden_layers=2, num_attention_heads=2,
                                     image_size=128, patch_size=16, window_size=3, global_attn_indexes=[1], mlp_dim=128,
                                     num_pos_feats=32)
    p = transformers.SamPromptEncoderConfig(hidden_size=64, image_size=128, patch_size=16, image_embedding_size=8)
    m = transformers.SamMaskDecoderConfig(hidden_size=64, num_attention_heads=2, mlp_dim=128, iou_head_hidden_dim=64,
                                          num_hidden_layers=2)
    return transformers.SamConfig(vision_config=v, prompt_encoder_config=p, mask_decoder_config=m)


def build_hf(transformers, cfg, seed=0):
    cfg._attn_implementation = "eager"
    model = transformers.SamModel(cfg).eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith("positional_embedding"):
                prm.copy_(torch.randn(prm.shape, generator=g))                       # Gaussian Fourier matrix, scale 1
            elif prm.dim() == 1 and name.endswith(".weight"):
                prm.copy_(1.0 + 0.1 * torch.randn(prm.shape, generator=g))           # norm gains
            elif name.endswith(".bias"):
                prm.copy_(0.1 * torch.randn(prm.shape, generator=g))
            elif "rel_pos" in name:
                prm.copy_(0.3 * torch.randn(prm.shape, generator=g))
            elif "pos_embed" in name or "embed" in name or "token" in name:
                prm.copy_(0.5 * torch.randn(prm.shape, generator=g))
            else:
                fan_in = prm[0].numel() if "upscale_conv" not in name else prm.shape[0]
                prm.copy_(torch.randn(prm.shape, generator=g) * fan_in ** -0.5)
            if ".q_proj." in name or ".k_proj." in name:
                # unit-variance q and k give the 4096-token decoder attentions logits of several hundred (one-hot
                # softmaxes that amplify any rounding); a trained SAM is nowhere near that
                prm.mul_(0.25)
        # tied in the Hugging Face model (prompt encoder shares the image-wide Fourier matrix)
        model.prompt_encoder.shared_embedding.positional_embedding.copy_(model.shared_image_embedding.positional_embedding)
    return model


def inputs(cfg, B=1, P=2, seed=1, points=False):
    g = torch.Generator().manual_seed(seed)
    S = cfg.vision_config.image_size
    px = torch.randn(B, 3, S, S, generator=g)
    lo = torch.rand(B, P, 2, generator=g) * S * 0.5
    hi = lo + 8 + torch.rand(B, P, 2, generator=g) * S * 0.4
    out = dict(pixel_values=px)
    if points:
        out["input_points"] = (lo + (hi - lo) * 0.5).reshape(B, P, 1, 2)
    else:
        out["input_boxes"] = torch.cat([lo, hi], dim=-1)
    return out


def refine_config(transformers):
    """facebook/sam-vit-base geometry everywhere the processor and the decoder see it (1024^2 input, 64x64 tokens,
    14x14 windows, 256-channel embeddings, default prompt encoder / mask decoder) with a 2-block, 64-wide vision tower
    so that the reference's CPU run of it (oracle/make_golden_sam.py) takes seconds."""
    v = transformers.SamVisionConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, global_attn_indexes=[1],
                                     mlp_dim=128)
    return transformers.SamConfig(vision_config=v)


def build_refine_hf(transformers, seed=0):
    """build_hf(refine_config) with the mask logits shifted down by a different constant per mask token: channel 0 of the
    upscaled embedding is made the constant GELU(4) and every hyper-network emits a fixed negative weight for it.  With
    plain random parameters the logits are sign-symmetric noise, every candidate covers the whole latent grid after the
    reference's `interpolate(...).bool()`, and the selection rules would have nothing to select."""
    model = build_hf(transformers, refine_config(transformers), seed)
    md = model.mask_decoder
    with torch.no_grad():
        md.upscale_conv2.weight[:, 0].zero_()
        md.upscale_conv2.bias[0] = 4.0
        for i, mlp in enumerate(md.output_hypernetworks_mlps):
            mlp.proj_out.weight[0].zero_()
            mlp.proj_out.bias[0] = -0.25 * (0.2 + 0.15 * i)
    return model


def refine_inputs():
    """Two synthetic 512x512 uint8 images (smooth blobs) with layout boxes (x0, y0, x1, y1 proportions) and a smooth
    64x64 'cross-attention' map per box."""
    import numpy as np
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(7)
    low = torch.rand(2, 3, 16, 16, generator=g)
    imgs = (F.interpolate(low, size=(512, 512), mode="bicubic", align_corners=False).clamp(0, 1) * 255).round().byte()
    images = [im.permute(1, 2, 0).numpy() for im in imgs]
    boxes = [[(0.10, 0.20, 0.55, 0.70), (0.50, 0.35, 0.95, 0.90)], [(0.25, 0.10, 0.80, 0.60), (0.05, 0.55, 0.45, 0.98)]]
    attn = []
    yy, xx = np.mgrid[0:64, 0:64] / 64.0
    for per_image in boxes:
        for (x0, y0, x1, y1) in per_image:
            cx, cy, sx, sy = (x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0) / 3, (y1 - y0) / 3
            attn.append(np.exp(-((xx - cx) / sx) ** 2 - ((yy - cy) / sy) ** 2).astype(np.float32))
    return images, boxes, attn
