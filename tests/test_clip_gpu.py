"""CLIP text encoder on the HIP kernels vs the Hugging Face `CLIPTextModel` the reference
calls (models/models.py:63-89, models/pipelines.py:303-304), random-initialised at the SD1.x text-tower size
(no checkpoints in the sandbox), fp32 on the CPU.  Tolerance: fp16 compute vs fp32, relative to the tensor max."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import lgd_amd  # noqa: E402,F401
from lgd_amd import ops  # noqa: E402
from lgd_amd.clip import CLIPTextConfig, HipCLIPTextEncoder  # noqa: E402


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_causal_attention_kernel(dev):
    B, H, S, d = 3, 12, 77, 64
    C = H * d
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, S, 3 * C, generator=g).to(dev).half()
    o = torch.empty(B * S, C, device=dev, dtype=torch.float16)
    q2 = qkv.reshape(B * S, 3 * C)
    ops.attn_causal_fwd(q2, q2[:, C:], q2[:, 2 * C:], o, B, H, S, d, d ** -0.5, view=(3 * C, S * 3 * C))
    sp = lambda t: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    q, k, v = sp(qkv[..., :C]), sp(qkv[..., C:2 * C]), sp(qkv[..., 2 * C:])
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * d ** -0.5
    s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=dev), 1), float("-inf"))
    ref = torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), v).permute(0, 2, 1, 3).reshape(B * S, C)
    assert relerr(o, ref) < 4e-3


def test_clip_text_encoder_vs_transformers(dev):
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                         num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                                         eos_token_id=2, bos_token_id=0, pad_token_id=1)
    torch.manual_seed(0)
    hf = transformers.CLIPTextModel(hf_cfg).float().eval()
    enc = HipCLIPTextEncoder(CLIPTextConfig(), hf.state_dict(), dev)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 40000, (3, 77), generator=g)
    ids[:, 0] = 49406
    for b, n in enumerate((9, 30, 76)):                  # EOS, then padding with the EOS id (SD tokenizer convention)
        ids[b, n:] = 49407
    with torch.no_grad():
        ref = hf(input_ids=ids)
    out = enc(ids)
    torch.cuda.synchronize()
    e_h, e_p = relerr(out[0], ref[0]), relerr(out.pooler_output, ref.pooler_output)
    print(f"CLIP text encoder: hidden relerr {e_h:.3e}, pooled relerr {e_p:.3e}")
    assert out[0].shape == (3, 77, 768) and out.pooler_output.shape == (3, 768)
    assert e_h < 2e-2 and e_p < 2e-2
    # unpadded phrases of different lengths, as prepare_gligen_condition batches them (padding=True)
    short = enc(ids[:, :32])
    assert relerr(short[0], ref[0][:, :32]) < 2e-2        # causal: a prefix does not depend on what follows


def test_clip_text_encoder_sd2_tower_vs_transformers(dev):
    """SD2.x text tower (OpenCLIP ViT-H/14 as exported for stable-diffusion-2: hidden 1024, 23 layers, 16 heads, exact
    GELU; BASELINE config 3 uses it for its prompts), random init, vs transformers' CLIPTextModel in fp32."""
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.CLIPTextConfig(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                                         num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu",
                                         eos_token_id=2, bos_token_id=0, pad_token_id=1)
    torch.manual_seed(0)
    hf = transformers.CLIPTextModel(hf_cfg).float().eval()
    cfg = CLIPTextConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                         hidden_act="gelu")
    from lgd_amd.clip import from_hf
    enc = from_hf(hf, dev)                                  # what models.load_sd does with the checkpoint's text tower
    assert enc.cfg == cfg
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 40000, (2, 77), generator=g)
    ids[:, 0] = 49406
    for b, n in enumerate((12, 70)):
        ids[b, n:] = 49407
    with torch.no_grad():
        ref = hf(input_ids=ids)
    out = enc(ids)
    torch.cuda.synchronize()
    e_h, e_p = relerr(out[0], ref[0]), relerr(out.pooler_output, ref.pooler_output)
    print(f"SD2 text tower: hidden relerr {e_h:.3e}, pooled relerr {e_p:.3e}")
    assert out[0].shape == (2, 77, 1024) and e_h < 2e-2 and e_p < 2e-2
    with pytest.raises(RuntimeError):
        HipCLIPTextEncoder(CLIPTextConfig(hidden_act="relu"), hf.state_dict(), dev)
