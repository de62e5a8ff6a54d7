"""CLIP text encoder on the HIP kernels.

The reference encodes every prompt with the Hugging Face `CLIPTextModel` ([ext] transformers 4.29.2):
models/models.py:63-89 (`encode_prompts`: per-box prompts, overall prompt, negative prompts -> 77 x 768 hidden
states) and models/pipelines.py:303-304 (GLIGEN phrase embeddings = `pooler_output`).  This module runs the
same network — token + position embeddings, 12 pre-LN transformer layers with CAUSAL self-attention and a
quick-GELU MLP, final LayerNorm, pooled state at the EOS token — on the C-ABI kernels (fused QKV GEMM, exact
two-pass causal attention, LayerNorm, GEMMs with bias / residual epilogues), fp16 with fp32 accumulation.

`HipCLIPTextEncoder(config, state_dict)` takes the parameter names of `CLIPTextModel.state_dict()` (with or
without the `text_model.` prefix) and is callable like the Hugging Face module as far as the reference uses it:
`enc(input_ids)[0]` (hidden states) and `.pooler_output`; it therefore drops into `model_dict.text_encoder`.
Tokenisation stays with the tokenizer object (host string processing).
"""
from dataclasses import dataclass

import torch

from . import ops

F16, F32 = torch.float16, torch.float32


@dataclass(frozen=True)
class CLIPTextConfig:
    """The fields of transformers' CLIPTextConfig that shape the computation (defaults = SD1.x: openai/clip-vit-large-patch14
    text tower; SD2.x = OpenCLIP ViT-H/14: hidden 1024, 23 layers, 16 heads, intermediate 4096, hidden_act "gelu")."""
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"
    eos_token_id: int = 2            # 2 = the legacy configs: pooled state taken at argmax(input_ids)


class _Out(tuple):
    pass


class HipCLIPTextEncoder:
    def __init__(self, config: CLIPTextConfig, state_dict, device="cuda"):
        if config.hidden_act not in ("quick_gelu", "gelu"):
            raise RuntimeError(f"hidden_act={config.hidden_act!r}: quick_gelu (SD1.x, CLIP ViT-L/14) and gelu (SD2.x, "
                               "OpenCLIP ViT-H/14) text towers are implemented")
        if config.hidden_size % config.num_attention_heads or (config.hidden_size // config.num_attention_heads) % 8:
            raise RuntimeError("head width must be a multiple of 8")
        self.cfg = config
        self.dev = torch.device(device)
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()}
        h16 = lambda t: t.detach().to(self.dev, F16).contiguous()
        f32 = lambda t: t.detach().to(self.dev, F32).contiguous()
        self.tok_emb = f32(sd["embeddings.token_embedding.weight"])
        self.pos_emb = f32(sd["embeddings.position_embedding.weight"])
        self.layers = []
        for i in range(config.num_hidden_layers):
            p = f"encoder.layers.{i}"
            qkv_w = torch.cat([sd[f"{p}.self_attn.{n}_proj.weight"] for n in "qkv"])
            qkv_b = torch.cat([sd[f"{p}.self_attn.{n}_proj.bias"] for n in "qkv"])
            self.layers.append(dict(
                ln1=(f32(sd[f"{p}.layer_norm1.weight"]), f32(sd[f"{p}.layer_norm1.bias"])),
                qkv=(h16(qkv_w), f32(qkv_b)),
                out=(h16(sd[f"{p}.self_attn.out_proj.weight"]), f32(sd[f"{p}.self_attn.out_proj.bias"])),
                ln2=(f32(sd[f"{p}.layer_norm2.weight"]), f32(sd[f"{p}.layer_norm2.bias"])),
                fc1=(h16(sd[f"{p}.mlp.fc1.weight"]), f32(sd[f"{p}.mlp.fc1.bias"])),
                fc2=(h16(sd[f"{p}.mlp.fc2.weight"]), f32(sd[f"{p}.mlp.fc2.bias"]))))
        self.ln_f = (f32(sd["final_layer_norm.weight"]), f32(sd["final_layer_norm.bias"]))
        # CLIPTextModelWithProjection (SDXL's text_encoder_2, OpenCLIP ViT-bigG/14): text_embeds = pooled @ W^T, no bias
        self.text_projection = h16(state_dict["text_projection.weight"]) if "text_projection.weight" in state_dict else None

    def to(self, *_a, **_k):            # call-surface compatibility with nn.Module users
        return self

    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, output_hidden_states=False, **_kw):
        """input_ids (B, S<=77) int64 -> (last_hidden_state (B,S,C) fp32,) with `.pooler_output` (B,C).
        output_hidden_states: `.hidden_states` = (embeddings, layer 1 output, ..., layer N output) as transformers
        returns them (before the final LayerNorm) — SDXL conditions on hidden_states[-2].  With a text_projection
        (CLIPTextModelWithProjection) `.text_embeds` = projected pooled state and element 0 of the output is
        text_embeds, as in transformers.
        Like the reference's call sites, no padding mask is applied (models/models.py:73-78 pass ids only; CLIP's
        text tower is causal, so a token never sees the padding behind it)."""
        cfg = self.cfg
        ids = input_ids.to(self.dev)
        B, S = ids.shape
        if S > cfg.max_position_embeddings:
            raise RuntimeError(f"{S} tokens exceed the text tower's {cfg.max_position_embeddings} positions")
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).bool().all()):
            raise NotImplementedError("padding masks are not applied (the reference passes ids only, models/models.py:73-78)")
        C, H = cfg.hidden_size, cfg.num_attention_heads
        d = C // H
        x = (self.tok_emb[ids] + self.pos_emb[:S].unsqueeze(0)).reshape(B * S, C).to(F16).contiguous()
        eps = cfg.layer_norm_eps
        hidden = [x.float().reshape(B, S, C)] if output_hidden_states else None
        for L in self.layers:
            h = ops.layernorm(x, L["ln1"][0], L["ln1"][1], eps)
            qkv = ops.linear(h, L["qkv"][0], L["qkv"][1])                          # [B*S, 3C]
            o = torch.empty((B * S, C), device=self.dev, dtype=F16)
            ops.attn_causal_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], o, B, H, S, d, d ** -0.5, view=(3 * C, S * 3 * C))
            x = ops.linear(o, L["out"][0], L["out"][1], res=x)
            h = ops.layernorm(x, L["ln2"][0], L["ln2"][1], eps)
            h = ops.linear(h, L["fc1"][0], L["fc1"][1])
            h = ops.quick_gelu(h) if cfg.hidden_act == "quick_gelu" else ops.act(h, ops.ACT_GELU)
            x = ops.linear(h, L["fc2"][0], L["fc2"][1], res=x)
            if hidden is not None:
                hidden.append(x.float().reshape(B, S, C))
        y = ops.layernorm(x, self.ln_f[0], self.ln_f[1], eps).float().reshape(B, S, C)
        if cfg.eos_token_id == 2:
            eos = ids.argmax(dim=-1)
        else:
            eos = (ids == cfg.eos_token_id).int().argmax(dim=-1)
        pooled = y[torch.arange(B, device=self.dev), eos]
        if self.text_projection is not None:
            te = ops.linear(pooled.to(F16).contiguous(), self.text_projection, None, out_f32=True)[:B]
            out = _Out((te, y))
            out.text_embeds = te
        else:
            out = _Out((y,))
        out.last_hidden_state = y
        out.pooler_output = pooled
        out.hidden_states = tuple(hidden) if hidden is not None else None
        return out


def from_hf(hf_text_model, device="cuda"):
    """A loaded transformers `CLIPTextModel` (any checkpoint of the SD1.x / SD2.x families) or `CLIPTextModelWithProjection`
    (SDXL text_encoder_2) -> the HIP encoder with the same call surface; what `models.load_sd` puts into
    `model_dict.text_encoder`."""
    c = hf_text_model.config
    cfg = CLIPTextConfig(vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                         num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                         max_position_embeddings=c.max_position_embeddings, layer_norm_eps=c.layer_norm_eps,
                         hidden_act=c.hidden_act, eos_token_id=c.eos_token_id)
    return HipCLIPTextEncoder(cfg, hf_text_model.state_dict(), device)
