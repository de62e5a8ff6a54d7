"""SAM (Segment Anything, ViT image encoder + prompt encoder + mask decoder) on the HIP kernels — SURVEY.md §8f rank 2.

The reference refines every per-box foreground mask with the Hugging Face `SamModel` ([ext] transformers 4.29.2,
facebook/sam-vit-base): models/sam.py:25-55 (`sam`: processor -> `sam_model(**inputs)` -> `post_process_masks`),
called from :125-172 (`sam_refine_attn`, LMD) and :182-213 (`sam_refine_boxes`, LMD+).  This module runs the same
network on the C-ABI kernels, fp16 with fp32 accumulation (the reference runs it under `torch.autocast`):

  image encoder   patch embedding as one GEMM over 16x16 patches (+ absolute position table in the epilogue);
                  12 pre-LN blocks: fused QKV GEMM -> `lgd_sam_relpos_qkv_f16` (window partition with SAM's
                  post-LayerNorm zero padding + decomposed relative-position bias folded into the head dimension,
                  csrc/sam.hip) -> flash attention (`lgd_attn_fwd_f16`, d = 96 for the 14x14 windows, 192 for the four
                  global blocks) -> `lgd_sam_window_merge_f16` -> projection GEMM with residual epilogue;
                  GEMM -> exact GELU -> GEMM(+residual) MLP; neck = 1x1 GEMM, channel LayerNorm, implicit-GEMM 3x3
                  conv, channel LayerNorm (channels-last throughout, so "LayerNorm2d" is the row LayerNorm kernel)
  prompt encoder  random-Fourier position encoding of <= a few points per prompt (host-sized arithmetic in torch)
  mask decoder    two-way transformer on [iou, 4 mask, prompt] tokens x 4096 image tokens: every attention through
                  `lgd_attn_fwd_f16` (d = 32 / 16), `keys + position` folded into the K/Q projections' residual
                  epilogue (the projected position table is precomputed at load), the two stride-2 transposed
                  convolutions as GEMMs over (dy, dx, c_out) columns, hyper-network MLPs, mask logits as one
                  K = 32 GEMM per prompt with fp32 output.

`HipSamModel(config, state_dict)` takes the parameter names of `SamModel.state_dict()` and is callable the way the
reference calls the Hugging Face module: `model(pixel_values=..., input_boxes=... | input_points=...,
**processor_extras)` -> object with `.pred_masks` (B, P, 3, 256, 256) fp32 logits and `.iou_scores` (B, P, 3); it
therefore drops into `sam_model_dict["sam_model"]`.  Image resizing / normalisation and `post_process_masks` stay
with the processor object (host preprocessing), as tokenisation does for the text encoder.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import ops
from .weightstore import pack_conv

F16, F32 = torch.float16, torch.float32


@dataclass(frozen=True)
class SamConfig:
    """The fields of transformers' SamConfig that shape the computation (defaults: facebook/sam-vit-base)."""
    image_size: int = 1024
    patch_size: int = 16
    num_channels: int = 3
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    mlp_dim: int = 3072
    window_size: int = 14
    global_attn_indexes: tuple = (2, 5, 8, 11)
    output_channels: int = 256
    vision_eps: float = 1e-6
    dec_hidden: int = 256
    dec_layers: int = 2
    dec_heads: int = 8
    dec_mlp_dim: int = 2048
    attention_downsample_rate: int = 2
    num_multimask_outputs: int = 3
    dec_eps: float = 1e-6

    @classmethod
    def from_hf(cls, c):
        v, m = c.vision_config, c.mask_decoder_config
        if v.hidden_act != "gelu" or m.hidden_act != "relu" or not v.use_rel_pos or not v.use_abs_pos or not v.qkv_bias:
            raise RuntimeError("only the SAM ViT layout of facebook/sam-vit-* is implemented")
        return cls(image_size=v.image_size, patch_size=v.patch_size, num_channels=v.num_channels,
                   hidden_size=v.hidden_size, num_hidden_layers=v.num_hidden_layers,
                   num_attention_heads=v.num_attention_heads, mlp_dim=v.mlp_dim, window_size=v.window_size,
                   global_attn_indexes=tuple(v.global_attn_indexes), output_channels=v.output_channels,
                   vision_eps=v.layer_norm_eps, dec_hidden=m.hidden_size, dec_layers=m.num_hidden_layers,
                   dec_heads=m.num_attention_heads, dec_mlp_dim=m.mlp_dim,
                   attention_downsample_rate=m.attention_downsample_rate,
                   num_multimask_outputs=m.num_multimask_outputs, dec_eps=m.layer_norm_eps)


class SamOutput:
    def __init__(self, iou_scores, pred_masks):
        self.iou_scores, self.pred_masks = iou_scores, pred_masks

    def __getitem__(self, i):
        return (self.iou_scores, self.pred_masks)[i]


def _rel_table(rel_pos, size):
    """[ext] SamVisionAttention.get_rel_pos for q_size == k_size: the (2*size-1, d) table, linearly resized when the
    checkpoint was trained at another grid size (load-time host work)."""
    L = 2 * size - 1
    if rel_pos.shape[0] != L:
        rel_pos = F.interpolate(rel_pos.float().reshape(1, rel_pos.shape[0], -1).transpose(1, 2), size=L,
                                mode="linear").reshape(-1, L).permute(1, 0)
    return rel_pos


class HipSamModel:
    def __init__(self, config: SamConfig, state_dict, device="cuda"):
        cfg = self.cfg = config
        self.dev = dev = torch.device(device)
        sd = state_dict
        h16 = lambda t: t.detach().to(dev, F16).contiguous()
        f32 = lambda t: t.detach().to(dev, F32).contiguous()
        lin = lambda p: (h16(sd[p + ".weight"]), f32(sd[p + ".bias"]))
        ln = lambda p: (f32(sd[p + ".weight"]), f32(sd[p + ".bias"]))
        self.grid = g = cfg.image_size // cfg.patch_size
        C, NH = cfg.hidden_size, cfg.num_attention_heads
        if C % NH or (C // NH) % 8:
            raise RuntimeError("head width must be a multiple of 8")
        self.d = d = C // NH
        if (cfg.num_channels * cfg.patch_size ** 2) % 8:
            raise RuntimeError("patch vector length must be a multiple of 8")

        # ---- image encoder
        v = "vision_encoder."
        self.patch = (h16(sd[v + "patch_embed.projection.weight"].reshape(C, -1)), f32(sd[v + "patch_embed.projection.bias"]))
        self.pos = h16(sd[v + "pos_embed"].reshape(g * g, C))
        self._pos_rep = {}
        self.blocks = []
        for i in range(cfg.num_hidden_layers):
            p = f"{v}layers.{i}."
            window = 0 if i in cfg.global_attn_indexes else cfg.window_size
            S = window or g
            DA = -(-(d + 2 * S) // 32) * 32                      # augmented head width (32-column MFMA chunks)
            if DA > 192:
                raise RuntimeError(f"augmented head width {DA} > 192 (head {d} + 2 x {S} bias columns)")
            self.blocks.append(dict(
                window=window, S=S, DA=DA, ln1=ln(p + "layer_norm1"), qkv=lin(p + "attn.qkv"),
                rel_h=f32(_rel_table(sd[p + "attn.rel_pos_h"], S)), rel_w=f32(_rel_table(sd[p + "attn.rel_pos_w"], S)),
                proj=lin(p + "attn.proj"), ln2=ln(p + "layer_norm2"), lin1=lin(p + "mlp.lin1"), lin2=lin(p + "mlp.lin2")))
        OC = cfg.output_channels
        self.neck1 = h16(sd[v + "neck.conv1.weight"].reshape(OC, C))
        self.neck_ln1 = ln(v + "neck.layer_norm1")
        self.neck2 = h16(pack_conv(sd[v + "neck.conv2.weight"]))
        self.neck_ln2 = ln(v + "neck.layer_norm2")

        # ---- prompt encoder (fp32, host-sized)
        key = "shared_image_embedding.positional_embedding"
        self.gauss = f32(sd[key] if key in sd else sd["prompt_encoder.shared_embedding.positional_embedding"])
        self.point_embed = [f32(sd[f"prompt_encoder.point_embed.{i}.weight"])[0] for i in range(4)]
        self.not_a_point = f32(sd["prompt_encoder.not_a_point_embed.weight"])[0]
        no_mask = f32(sd["prompt_encoder.no_mask_embed.weight"])[0]
        # `image_embeddings + dense_prompt_embeddings` with no mask prompt is a per-channel constant: folded into the
        # shift of the neck's last LayerNorm
        self.neck_ln2_dense = (self.neck_ln2[0], self.neck_ln2[1] + no_mask)
        ax = (torch.arange(g, device=dev, dtype=F32) + 0.5) / g
        yy, xx = torch.meshgrid(ax, ax, indexing="ij")
        self.image_pe = self._pe(torch.stack([xx, yy], dim=-1).reshape(g * g, 2))           # [g*g, D] fp32

        # ---- mask decoder
        m = "mask_decoder."
        D = cfg.dec_hidden
        if D != OC:
            raise RuntimeError("mask decoder width must equal the encoder's output channels")
        self.out_tokens = torch.cat([f32(sd[m + "iou_token.weight"]), f32(sd[m + "mask_tokens.weight"])])   # [1+nm, D]
        self.n_mask_tokens = cfg.num_multimask_outputs + 1

        def attn(p):
            q, k, vv, o = (lin(f"{p}.{n}_proj") for n in ("q", "k", "v", "out"))
            return dict(q=q, k=k, v=vv, o=o, inner=q[0].shape[0],
                        kv=(torch.cat([k[0], vv[0]]).contiguous(), torch.cat([k[1], vv[1]]).contiguous()),
                        qk=(torch.cat([q[0], k[0]]).contiguous(), torch.cat([q[1], k[1]]).contiguous()),
                        qkv=(torch.cat([q[0], k[0], vv[0]]).contiguous(), torch.cat([q[1], k[1], vv[1]]).contiguous()))

        def with_pe(a, which):
            """`(keys + image_pe) @ W^T` = `keys @ W^T + image_pe @ W^T`: the second term is a constant table, added in
            the projection's residual epilogue.  which="kv": [pe@Wk^T | 0] for the fused K/V GEMM; "q": pe@Wq^T."""
            w = a[which[0]][0].float()
            t = self.image_pe @ w.t()
            if which == "kv":
                t = torch.cat([t, torch.zeros_like(t)], dim=1)
            return t.to(F16).contiguous()

        self.dec = []
        for i in range(cfg.dec_layers):
            p = f"{m}transformer.layers.{i}."
            t2i, i2t = attn(p + "cross_attn_token_to_image"), attn(p + "cross_attn_image_to_token")
            self.dec.append(dict(self_attn=attn(p + "self_attn"), ln1=ln(p + "layer_norm1"), t2i=t2i,
                                 t2i_pe=with_pe(t2i, "kv"), ln2=ln(p + "layer_norm2"), lin1=lin(p + "mlp.lin1"),
                                 lin2=lin(p + "mlp.lin2"), ln3=ln(p + "layer_norm3"), ln4=ln(p + "layer_norm4"),
                                 i2t=i2t, i2t_pe=with_pe(i2t, "q")))
        self.final = attn(m + "transformer.final_attn_token_to_image")
        self.final_pe = with_pe(self.final, "kv")
        self.ln_final = ln(m + "transformer.layer_norm_final_attn")
        up1, up2 = sd[m + "upscale_conv1.weight"], sd[m + "upscale_conv2.weight"]     # ConvTranspose2d: [Cin, Cout, 2, 2]
        # kernel 2, stride 2: every input pixel produces its own 2x2 output block -> GEMM over (dy, dx, c_out) columns
        self.up1 = (h16(up1.permute(2, 3, 1, 0).reshape(-1, up1.shape[0])), f32(sd[m + "upscale_conv1.bias"].repeat(4)))
        self.up2 = (h16(up2.permute(2, 3, 1, 0).reshape(-1, up2.shape[0])), f32(sd[m + "upscale_conv2.bias"].repeat(4)))
        self.up_c1, self.up_c2 = up1.shape[1], up2.shape[1]
        self.up_ln = ln(m + "upscale_layer_norm")

        def ffn(p):
            n_mid = len({k for k in sd if k.startswith(p + ".layers.") and k.endswith(".weight")})
            return [lin(p + ".proj_in")] + [lin(f"{p}.layers.{j}") for j in range(n_mid)] + [lin(p + ".proj_out")]
        self.hyper = [ffn(f"{m}output_hypernetworks_mlps.{i}") for i in range(self.n_mask_tokens)]
        self.iou_head = ffn(m + "iou_prediction_head")
        self._pe_rep = {}

    def to(self, *_a, **_k):            # call-surface compatibility with nn.Module users
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------------------------------------------------
    def _pe(self, coords01):
        """[ext] SamPositionalEmbedding.forward: random-Fourier features of points in [0,1]^2 (x, y)."""
        c = (2.0 * coords01.to(self.dev, F32) - 1.0) @ self.gauss
        c = 2.0 * math.pi * c
        return torch.cat([c.sin(), c.cos()], dim=-1)

    def _rep(self, cache, key, t, n):
        k = (key, n)
        if k not in cache:
            if len(cache) > 16:
                cache.clear()
            cache[k] = t.repeat(n, 1).contiguous()
        return cache[k]

    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_image(self, pixel_values, _dense=False):
        """pixel_values (B, 3, S, S) -> image embeddings, channels-last tokens [B*g*g, output_channels] fp16
        ([ext] SamVisionEncoder.forward; the Hugging Face module returns the same values as (B, C, g, g))."""
        cfg, g, d = self.cfg, self.grid, self.d
        B, Cin, Hh, Ww = pixel_values.shape
        if (Cin, Hh, Ww) != (cfg.num_channels, cfg.image_size, cfg.image_size):
            raise ValueError(f"Input image size ({Hh}*{Ww}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")
        P, C, NH = cfg.patch_size, cfg.hidden_size, cfg.num_attention_heads
        patches = (pixel_values.to(self.dev, F16).reshape(B, Cin, g, P, g, P).permute(0, 2, 4, 1, 3, 5)
                   .reshape(B * g * g, Cin * P * P).contiguous())
        x = ops.linear(patches, self.patch[0], self.patch[1], res=self._rep(self._pos_rep, "pos", self.pos, B))
        scale = d ** -0.5
        for L in self.blocks:
            h = ops.layernorm(x, L["ln1"][0], L["ln1"][1], cfg.vision_eps)
            qkv = ops.linear(h, L["qkv"][0], L["qkv"][1])
            S, DA = L["S"], L["DA"]
            qa, ka, va = ops.sam_relpos_qkv(qkv, L["qkv"][1], L["rel_h"], L["rel_w"], B, g, g, L["window"], NH, d, DA, scale)
            nwin = qa.shape[0] // (S * S)
            oa = torch.empty_like(qa)
            ops.attn_fwd(qa, ka, va, oa, nwin, NH, S * S, S * S, DA, scale)
            o = ops.sam_window_merge(oa, B, g, g, L["window"], NH, d, DA)
            x = ops.linear(o, L["proj"][0], L["proj"][1], res=x)
            h = ops.layernorm(x, L["ln2"][0], L["ln2"][1], cfg.vision_eps)
            h = ops.act(ops.linear(h, L["lin1"][0], L["lin1"][1]), ops.ACT_GELU)
            x = ops.linear(h, L["lin2"][0], L["lin2"][1], res=x)
        y = ops.linear(x, self.neck1)
        y = ops.layernorm(y, self.neck_ln1[0], self.neck_ln1[1], 1e-6)
        y = ops.conv3x3(y, self.neck2, B, g, g)
        ln2 = self.neck_ln2_dense if _dense else self.neck_ln2
        return ops.layernorm(y, ln2[0], ln2[1], 1e-6)

    def get_image_embeddings(self, pixel_values, **_kw):
        """(B, C, g, g) fp32, as SamModel.get_image_embeddings."""
        e = self.encode_image(pixel_values)
        B = pixel_values.shape[0]
        return e.float().reshape(B, self.grid, self.grid, -1).permute(0, 3, 1, 2).contiguous()

    # ------------------------------------------------------------------------------------------------------------
    def _prompt_tokens(self, B, input_points, input_labels, input_boxes):
        """[ext] SamPromptEncoder.forward (sparse part) -> (B, P, T_s, D) fp32."""
        size = float(self.cfg.image_size)
        sparse = None
        if input_points is not None:
            pts = input_points.to(self.dev, F32)
            if pts.dim() != 4:
                raise ValueError("The input_points must be a 4D tensor. Of shape `batch_size`, `point_batch_size`, "
                                 "`nb_points_per_image`, `2`.")
            labels = (torch.ones(pts.shape[:3], device=self.dev, dtype=torch.int64) if input_labels is None
                      else input_labels.to(self.dev))
            pts = pts + 0.5
            if input_boxes is None:
                pts = torch.cat([pts, torch.zeros_like(pts[:, :, :1])], dim=2)
                labels = torch.cat([labels, -torch.ones_like(labels[:, :, :1])], dim=2)
            e = self._pe(pts / size)
            lab = labels[..., None]
            e = torch.where(lab == -1, self.not_a_point, e)
            e = torch.where(lab != -10, e, torch.zeros_like(e))
            e = torch.where(lab == 0, e + self.point_embed[0], e)
            e = torch.where(lab == 1, e + self.point_embed[1], e)
            sparse = e
        if input_boxes is not None:
            bx = input_boxes.to(self.dev, F32)
            if bx.dim() != 3:
                raise ValueError("The input_points must be a 3D tensor. Of shape `batch_size`, `nb_boxes`, `4`.")
            e = self._pe((bx + 0.5).reshape(bx.shape[0], bx.shape[1], 2, 2) / size).clone()
            e[:, :, 0] += self.point_embed[2]
            e[:, :, 1] += self.point_embed[3]
            sparse = e if sparse is None else torch.cat([sparse, e], dim=2)
        if sparse is not None and sparse.shape[0] != B:
            raise ValueError("The batch size of the image embeddings and the prompts must be the same.")
        return sparse

    def _attention(self, a, q_in, k_in, v_in, BP, Sq, Sk, *, k_pe=None, q_pe=None, res=None):
        """[ext] SamAttention.forward on flattened tokens: q_in [BP*Sq, D], k_in / v_in [BP*Sk, D].  Equal inputs share
        one fused projection GEMM; k_pe / q_pe are pre-projected position tables for the residual epilogue."""
        H = self.cfg.dec_heads
        inner = a["inner"]
        dh = inner // H
        if q_in is k_in and k_in is v_in:
            f = ops.linear(q_in, a["qkv"][0], a["qkv"][1])
            q, k, v = f, f[:, inner:], f[:, 2 * inner:]
            qv = kv = vv = (3 * inner, None)
        elif q_in is k_in:
            f = ops.linear(q_in, a["qk"][0], a["qk"][1])
            q, k, qv, kv = f, f[:, inner:], (2 * inner, None), (2 * inner, None)
            v, vv = ops.linear(v_in, a["v"][0], a["v"][1]), (inner, None)
        else:
            q, qv = ops.linear(q_in, a["q"][0], a["q"][1], res=q_pe), (inner, None)
            if k_pe is not None:                                   # k = (v_in + pe) Wk, v = v_in Wv in one GEMM
                f = ops.linear(v_in, a["kv"][0], a["kv"][1], res=k_pe)
                k, v, kv, vv = f, f[:, inner:], (2 * inner, None), (2 * inner, None)
            else:
                k, kv = ops.linear(k_in, a["k"][0], a["k"][1]), (inner, None)
                v, vv = ops.linear(v_in, a["v"][0], a["v"][1]), (inner, None)
        o = torch.empty((BP * Sq, inner), device=self.dev, dtype=F16)
        view = lambda ld, S: (ld[0], S * ld[0])
        ops.attn_fwd(q, k, v, o, BP, H, Sq, Sk, dh, dh ** -0.5, q_view=view(qv, Sq), k_view=view(kv, Sk),
                     v_view=view(vv, Sk))
        return ops.linear(o, a["o"][0], a["o"][1], res=res)

    def _ffn(self, layers, x, out_f32=False):
        """[ext] SamFeedForward: Linear/ReLU ... Linear."""
        for i, (w, b) in enumerate(layers):
            last = i == len(layers) - 1
            x = ops.linear(x, w, b, out_f32=out_f32 and last)
            if not last:
                x = ops.act(x, ops.ACT_RELU)
        return x

    @torch.no_grad()
    def decode_masks(self, emb_dense, sparse, multimask_output=True):
        """[ext] SamMaskDecoder.forward.  emb_dense [B*g*g, D] fp16 (image embeddings + dense prompt embedding),
        sparse (B, P, T_s, D) fp32 -> (pred_masks (B, P, n, 4g, 4g) fp32, iou_scores (B, P, n) fp32)."""
        cfg, g = self.cfg, self.grid
        D, HW = cfg.dec_hidden, g * g
        B, P, Ts, _ = sparse.shape
        BP = B * P
        nm = self.n_mask_tokens
        T = 1 + nm + Ts
        if (T * D) % 8:
            raise RuntimeError("token block must be a multiple of 8 halfs")
        tokens = torch.cat([self.out_tokens.expand(B, P, -1, -1), sparse], dim=2).reshape(BP * T, D)
        qpe = tokens.to(F16).contiguous()                       # query_point_embedding (fixed through the transformer)
        queries = qpe
        keys = (emb_dense.reshape(B, 1, HW, D).expand(B, P, HW, D).reshape(BP * HW, D).contiguous() if P > 1
                else emb_dense)
        eps = cfg.dec_eps
        rep = lambda name, t: self._rep(self._pe_rep, name, t, BP)
        for i, L in enumerate(self.dec):
            if i == 0:                                           # skip_first_layer_pe: the attention output REPLACES the tokens
                queries = self._attention(L["self_attn"], queries, queries, queries, BP, T, T)
            else:
                qp = ops.add(queries, qpe)
                queries = self._attention(L["self_attn"], qp, qp, queries, BP, T, T, res=queries)
            queries = ops.layernorm(queries, L["ln1"][0], L["ln1"][1], eps)
            qp = ops.add(queries, qpe)
            queries = self._attention(L["t2i"], qp, None, keys, BP, T, HW, k_pe=rep(f"t2i{i}", L["t2i_pe"]), res=queries)
            queries = ops.layernorm(queries, L["ln2"][0], L["ln2"][1], eps)
            h = ops.act(ops.linear(queries, L["lin1"][0], L["lin1"][1]), ops.ACT_RELU)
            queries = ops.linear(h, L["lin2"][0], L["lin2"][1], res=queries)
            queries = ops.layernorm(queries, L["ln3"][0], L["ln3"][1], eps)
            qp = ops.add(queries, qpe)
            keys = self._attention(L["i2t"], keys, qp, queries, BP, HW, T, q_pe=rep(f"i2t{i}", L["i2t_pe"]), res=keys)
            keys = ops.layernorm(keys, L["ln4"][0], L["ln4"][1], eps)
        qp = ops.add(queries, qpe)
        queries = self._attention(self.final, qp, None, keys, BP, T, HW, k_pe=rep("final", self.final_pe), res=queries)
        queries = ops.layernorm(queries, self.ln_final[0], self.ln_final[1], 1e-5)

        # upscaling: rows stay in (prompt, y, x[, dy1, dx1[, dy2, dx2]]) order; un-nested once at the end
        u = ops.linear(keys, self.up1[0], self.up1[1]).reshape(BP * HW * 4, self.up_c1)
        u = ops.act(ops.layernorm(u, self.up_ln[0], self.up_ln[1], 1e-6), ops.ACT_GELU)
        u = ops.act(ops.linear(u, self.up2[0], self.up2[1]), ops.ACT_GELU).reshape(BP, HW * 16, self.up_c2)
        q3 = queries.reshape(BP, T, D)
        hyper = torch.stack([self._ffn(self.hyper[i], q3[:, 1 + i]) for i in range(nm)], dim=1)        # [BP, nm, c2]
        nmp = -(-nm // 4) * 4                                     # GEMM N is a multiple of 4
        if nmp != nm:
            hyper = torch.cat([hyper, torch.zeros((BP, nmp - nm, self.up_c2), device=self.dev, dtype=F16)], dim=1)
        hyper = hyper.contiguous()
        masks = torch.empty((BP, HW * 16, nmp), device=self.dev, dtype=F32)
        for p in range(BP):
            ops.linear(u[p], hyper[p], out=masks[p], out_f32=True)
        masks = (masks.reshape(BP, g, g, 2, 2, 2, 2, nmp).permute(0, 7, 1, 3, 5, 2, 4, 6)
                 .reshape(B, P, nmp, 4 * g, 4 * g)[:, :, :nm])
        iou = self._ffn(self.iou_head, q3[:, 0], out_f32=True).reshape(B, P, -1)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        return masks[:, :, sl].contiguous(), iou[:, :, sl].contiguous()

    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, pixel_values=None, input_points=None, input_labels=None, input_boxes=None, input_masks=None,
                 image_embeddings=None, multimask_output=True, attention_similarity=None, target_embedding=None, **_kw):
        """[ext] SamModel.forward as models/sam.py:39-40 calls it (`sam_model(**processor_output)`; the processor's
        `original_sizes` / `reshaped_input_sizes` are accepted and ignored like the Hugging Face module does)."""
        if pixel_values is None:
            raise ValueError("pixel_values must be provided (precomputed image_embeddings are not taken)"
                             if image_embeddings is not None else "Either pixel_values or image_embeddings must be provided.")
        if input_masks is not None or attention_similarity is not None or target_embedding is not None:
            raise NotImplementedError("mask prompts / PerSAM inputs are not used by the reference (models/sam.py:39)")
        if input_points is None and input_boxes is None:
            raise NotImplementedError("a point or box prompt is required (models/sam.py:57-61)")
        if input_points is not None and input_boxes is not None and input_points.shape[1] != input_boxes.shape[1]:
            raise ValueError("You should provide as many bounding boxes as input points per box.")
        B = pixel_values.shape[0]
        sparse = self._prompt_tokens(B, input_points, input_labels, input_boxes)
        emb = self.encode_image(pixel_values, _dense=True)
        masks, iou = self.decode_masks(emb, sparse, multimask_output)
        return SamOutput(iou_scores=iou, pred_masks=masks)
