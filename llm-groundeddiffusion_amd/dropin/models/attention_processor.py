"""models/attention_processor.py of the reference: the `Attention` layer handle and the `AttnProcessor`
hook (attention_processor.py:26-293, :296-483), executing on the HIP kernels.

`AttnProcessor.__call__` keeps the reference's keyword surface:
    p(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
      return_attntion_probs=False [sic], attn_key=None, attn_process_fn=None, return_cond_ca_only=False,
      return_token_ca_only=None, offload_cross_attn_to_cpu=False, save_attn_to_dict=None, save_keys=None,
      enable_flash_attn=True) -> hidden_states | (hidden_states, probs)
and its side effect `save_attn_to_dict[tuple(attn_key)] = probs (B', heads, HW, T')`.
Inside the engine's static plans the same computation is issued without going through Python objects;
this class is the layer-level entry point (tests, custom loops)."""
import torch

from lgd_amd import ops


class Attention:
    """One attention layer of the UNet bound to the engine's packed weights (query_dim = inner_dim = C)."""

    def __init__(self, engine, prefix: str, heads: int, cross: bool):
        self.engine, self.prefix, self.heads, self.cross = engine, prefix, heads, cross
        w = engine.w.h
        if cross:
            self.to_q_w, self.kv_w = w[f"{prefix}.to_q.w"], w[f"{prefix}.kv.w"]
        else:
            self.qkv_w = w[f"{prefix}.qkv.w"]
        self.out_w, self.out_b = w[f"{prefix}.to_out.0.w"], engine.w.f[f"{prefix}.to_out.0.b"]
        self.inner_dim = self.out_w.shape[0]
        self.scale = (self.inner_dim // heads) ** -0.5
        self.processor = AttnProcessor()
        # attributes the reference's processor reads
        self.spatial_norm = self.group_norm = self.norm_cross = None
        self.residual_connection, self.rescale_output_factor = False, 1.0

    def set_processor(self, processor):
        if not isinstance(processor, AttnProcessor):
            raise NotImplementedError("only AttnProcessor (the reference's default) runs on the HIP path")
        self.processor = processor

    def __call__(self, hidden_states, encoder_hidden_states=None, attention_mask=None, return_attntion_probs=False,
                 **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, return_attntion_probs=return_attntion_probs,
                              **cross_attention_kwargs)

    forward = __call__


class AttnProcessor:
    def __call__(self, attn: Attention, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 return_attntion_probs=False, attn_key=None, attn_process_fn=None, return_cond_ca_only=False,
                 return_token_ca_only=None, offload_cross_attn_to_cpu=False, save_attn_to_dict=None, save_keys=None,
                 enable_flash_attn=True):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never used by LMD / LMD+ (always None)")
        if attn_process_fn is not None:
            raise NotImplementedError("attn_process_fn is never set by LMD / LMD+")
        cross = encoder_hidden_states is not None
        B, S, C = hidden_states.shape
        H, d = attn.heads, C // attn.heads
        x = hidden_states.reshape(B * S, C).to(torch.float16).contiguous()
        want_map = cross and (return_attntion_probs or (save_attn_to_dict is not None and
                                                        (save_keys is None or tuple(attn_key) in save_keys)))
        o = torch.empty((B, S, C), device=x.device, dtype=torch.float16)
        probs = None
        if cross:
            ctx = encoder_hidden_states.to(torch.float16)
            T = ctx.shape[1]
            q = ops.linear(x, attn.to_q_w)
            kv = ops.linear(ctx.reshape(B * T, -1).contiguous(), attn.kv_w).view(B, T, 2 * C)
            if want_map:
                probs = torch.zeros((B, H, S, T), device=x.device, dtype=torch.float32)
            ops.cross_attn_fwd(q, kv, kv[:, :, C:], o, B, H, S, T, d, attn.scale, probs=probs,
                               k_view=(2 * C, T * 2 * C), v_view=(2 * C, T * 2 * C))
        else:
            qkv = ops.linear(x, attn.qkv_w)
            view = (3 * C, S * 3 * C)
            ops.attn_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], o, B, H, S, S, d, attn.scale, q_view=view, k_view=view,
                         v_view=view)
        out = ops.linear(o.view(B * S, C), attn.out_w, attn.out_b).view(B, S, C).to(hidden_states.dtype)
        if probs is not None:
            p = probs.to(hidden_states.dtype)
            if return_token_ca_only is not None:                          # attention_processor.py:466-473
                p = p[:, :, :, return_token_ca_only:return_token_ca_only + 1] \
                    if isinstance(return_token_ca_only, int) else p[:, :, :, return_token_ca_only]
            if return_cond_ca_only:                                       # :474-476
                assert B % 2 == 0, f"Samples are not in pairs: {B} samples"
                p = p[B // 2:]
            if offload_cross_attn_to_cpu:
                p = p.cpu()
            if save_attn_to_dict is not None and (save_keys is None or tuple(attn_key) in save_keys):
                save_attn_to_dict[tuple(attn_key)] = p
            if return_attntion_probs:
                return out, p
        return out


AttentionProcessor = AttnProcessor
