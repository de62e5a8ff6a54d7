"""ORACLE TEST INFRASTRUCTURE — CPU fp32 restatement of the reference's stage-2 hot path.

Plain functional torch (fp32, NCHW, autograd for the guidance gradient).  It needs neither
/root/reference nor diffusers, so it travels to the GPU box where `-m gpu` tests compare the HIP
path with it.  It is pinned against the reference's own code run through oracle/ref_harness.py:
see oracle/make_golden.py and tests/test_oracle.py (fixtures in tests/golden/).

Every function cites the reference file:line it restates.  [ext] marks arithmetic that lives in
diffusers 0.18.0 (not under /root/reference): restated from its published behaviour — parity
UNPINNED at that boundary.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math
from collections.abc import Iterable

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_GUIDANCE_ATTN_KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]  # pipelines.py:14


# =================================================================================================
# UNet forward (models/unet_2d_condition.py:704-980)
# =================================================================================================
def timestep_embedding(t, dim):
    """[ext] Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) — unet_2d_condition.py:305,801."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    e = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(e), torch.sin(e)], dim=-1)


def resnet(sd, p, x, temb, groups, eps):
    """[ext] ResnetBlock2D.forward (call sites unet_2d_blocks.py:255,274,425,687,787)."""
    h = F.silu(F.group_norm(x, groups, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps))
    h = F.conv2d(h, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = h + F.linear(F.silu(temb), sd[f"{p}.time_emb_proj.weight"], sd[f"{p}.time_emb_proj.bias"])[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps))
    h = F.conv2d(h, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"])
    if _TAPS is not None:
        _TAPS[f"{p}.out"] = (x + h).detach()
    return x + h


_TAPS = None  # optional dict collecting named intermediate activations (debugging only)


def attention(sd, p, x, ctx, heads, hook=None):
    """Attention + AttnProcessor (attention_processor.py:426-483 slow path == :338-363 fast path
    numerically): to_q/to_k/to_v, softmax(scale q k^T), bmm with v, to_out[0].  `hook(probs)` receives
    the (B, heads, S, T) probabilities still attached to the autograd graph (:465,:479-480)."""
    B, S, C = x.shape
    ctx = x if ctx is None else ctx
    d = C // heads
    q = F.linear(x, sd[f"{p}.to_q.weight"]).reshape(B, S, heads, d).permute(0, 2, 1, 3)
    k = F.linear(ctx, sd[f"{p}.to_k.weight"]).reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    v = F.linear(ctx, sd[f"{p}.to_v.weight"]).reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    probs = (torch.matmul(q, k.transpose(-1, -2)) * d ** -0.5).softmax(dim=-1)  # :216-228
    if hook is not None:
        hook(probs)
    o = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, S, C)
    return F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def feed_forward(sd, p, x):
    """FeedForward with GEGLU (attention.py:286-289, 333-335): exact-erf GELU."""
    h, gate = F.linear(x, sd[f"{p}.net.0.proj.weight"], sd[f"{p}.net.0.proj.bias"]).chunk(2, dim=-1)
    return F.linear(h * F.gelu(gate), sd[f"{p}.net.2.weight"], sd[f"{p}.net.2.bias"])


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{p}.weight"], sd[f"{p}.bias"], 1e-5)


def fuser(sd, p, x, objs, heads):
    """GatedSelfAttentionDense.forward (attention.py:43-53)."""
    n_visual = x.shape[1]
    objs = F.linear(objs, sd[f"{p}.linear.weight"], sd[f"{p}.linear.bias"])
    h = layer_norm(sd, f"{p}.norm1", torch.cat([x, objs], dim=1))
    x = x + sd[f"{p}.alpha_attn"].tanh() * attention(sd, f"{p}.attn", h, None, heads)[:, :n_visual]
    if _TAPS is not None:
        _TAPS[p.replace(".transformer_blocks.0.fuser", "") + ".after_fuser_attn"] = x.detach()
    x = x + sd[f"{p}.alpha_dense"].tanh() * feed_forward(sd, f"{p}.ff", layer_norm(sd, f"{p}.norm2", x))
    return x


class _Stop(Exception):
    pass


def transformer(sd, p, x, ctx, heads, groups, key, st):
    """Transformer2DModel.forward continuous branch (transformer_2d.py:279-327) around one
    BasicTransformerBlock (attention.py:156-237)."""
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    w_in = sd[f"{p}.proj_in.weight"]
    if w_in.dim() == 4:
        h = F.conv2d(h, w_in, sd[f"{p}.proj_in.bias"]).permute(0, 2, 3, 1).reshape(B, H * W, C)
    else:
        h = F.linear(h.permute(0, 2, 3, 1).reshape(B, H * W, C), w_in, sd[f"{p}.proj_in.bias"])
    t = f"{p}.transformer_blocks.0"
    h = attention(sd, f"{t}.attn1", layer_norm(sd, f"{t}.norm1", h), None, heads) + h
    if _TAPS is not None:
        _TAPS[f"{p}.after_attn1"] = h.detach()
    if st["objs"] is not None and st["fuser_enabled"]:
        h = fuser(sd, f"{t}.fuser", h, st["objs"], heads)          # attention.py:198-200
        if _TAPS is not None:
            _TAPS[f"{p}.after_fuser"] = h.detach()
    hook = None
    if st["saved"] is not None and (st["save_keys"] is None or key in st["save_keys"]):
        def hook(probs, key=key):                                   # attention_processor.py:463-480
            tok = st["token_only"]
            if tok is not None:
                probs = probs[:, :, :, tok:tok + 1] if isinstance(tok, int) else probs[:, :, :, tok]
            if st["cond_only"]:
                probs = probs[probs.shape[0] // 2:]
            st["saved"][key] = probs
    h = attention(sd, f"{t}.attn2", layer_norm(sd, f"{t}.norm2", h), ctx, heads, hook) + h
    if _TAPS is not None:
        _TAPS[f"{p}.after_attn2"] = h.detach()
    if st["stop_after"] is not None and key == st["stop_after"]:
        raise _Stop()
    h = feed_forward(sd, f"{t}.ff", layer_norm(sd, f"{t}.norm3", h)) + h
    w_out = sd[f"{p}.proj_out.weight"]
    if w_out.dim() == 4:
        h = F.conv2d(h.reshape(B, H, W, C).permute(0, 3, 1, 2), w_out, sd[f"{p}.proj_out.bias"])
    else:
        h = F.linear(h, w_out, sd[f"{p}.proj_out.bias"]).reshape(B, H, W, C).permute(0, 3, 1, 2)
    if _TAPS is not None:
        _TAPS[f"{p}.out"] = (h + res).detach()
    return h + res


def position_net(sd, boxes, masks, positive_embeddings):
    """PositionNet.forward + FourierEmbedder (unet_2d_condition.py:63-114)."""
    masks = masks.unsqueeze(-1)
    freq = 100 ** (torch.arange(8) / 8)
    x = freq[None, None, None] * boxes.unsqueeze(-1)
    xyxy = torch.stack((x.sin(), x.cos()), dim=-1).permute(0, 1, 3, 4, 2).reshape(*boxes.shape[:2], -1)
    pos = positive_embeddings * masks + (1 - masks) * sd["position_net.null_positive_feature"].view(1, 1, -1)
    xyxy = xyxy * masks + (1 - masks) * sd["position_net.null_position_feature"].view(1, 1, -1)
    h = torch.cat([pos, xyxy], dim=-1)
    h = F.silu(F.linear(h, sd["position_net.linears.0.weight"], sd["position_net.linears.0.bias"]))
    h = F.silu(F.linear(h, sd["position_net.linears.2.weight"], sd["position_net.linears.2.bias"]))
    return F.linear(h, sd["position_net.linears.4.weight"], sd["position_net.linears.4.bias"])


def unet_forward(sd, cfg, sample, t, ehs, *, saved=None, save_keys=None, token_only=None,
                 cond_only=False, gligen=None, fuser_enabled=True, stop_after=None, taps=None):
    """UNet2DConditionModel.forward (unet_2d_condition.py:704-980) for the SD 1.x/2.x configs.

    cfg: dict-like with block_out_channels, layers_per_block, attention_head_dim, norm_num_groups,
    norm_eps.  gligen: dict(boxes, masks, positive_embeddings) or None (:863-872).
    stop_after: attn key after whose cross-attention the forward is abandoned (returns None) — the
    algorithmic minimum for the guidance pass (TODO at pipelines.py:46); loss/grad are identical.
    """
    global _TAPS
    _TAPS = taps
    boc = list(cfg["block_out_channels"])
    heads_l = list(cfg["attention_head_dim"])
    groups, eps, lpb = cfg["norm_num_groups"], cfg["norm_eps"], cfg["layers_per_block"]
    n = len(boc)
    B = sample.shape[0]
    tt = torch.as_tensor(t).reshape(-1).expand(B)
    emb = timestep_embedding(tt, boc[0])
    emb = F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    st = dict(saved=saved, save_keys=[tuple(k) for k in save_keys] if save_keys is not None else None,
              token_only=token_only, cond_only=cond_only, fuser_enabled=fuser_enabled,
              stop_after=tuple(stop_after) if stop_after is not None else None, objs=None)
    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    if gligen is not None:
        st["objs"] = position_net(sd, gligen["boxes"], gligen["masks"], gligen["positive_embeddings"])
    try:
        skips = [x]
        for i in range(n):
            for j in range(lpb):
                x = resnet(sd, f"down_blocks.{i}.resnets.{j}", x, emb, groups, eps)
                if i < n - 1:
                    x = transformer(sd, f"down_blocks.{i}.attentions.{j}", x, ehs, heads_l[i], groups,
                                    ("down", i, j, 0), st)
                skips.append(x)
            if i < n - 1:
                p = f"down_blocks.{i}.downsamplers.0.conv"
                x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], stride=2, padding=1)  # [ext] Downsample2D
                skips.append(x)
        x = resnet(sd, "mid_block.resnets.0", x, emb, groups, eps)
        x = transformer(sd, "mid_block.attentions.0", x, ehs, heads_l[-1], groups, ("mid", 0, 0, 0), st)
        x = resnet(sd, "mid_block.resnets.1", x, emb, groups, eps)
        rev_heads = heads_l[::-1]
        for i in range(n):
            for j in range(lpb + 1):
                x = torch.cat([x, skips.pop()], dim=1)                               # unet_2d_blocks.py:646-649
                x = resnet(sd, f"up_blocks.{i}.resnets.{j}", x, emb, groups, eps)
                if i > 0:
                    x = transformer(sd, f"up_blocks.{i}.attentions.{j}", x, ehs, rev_heads[i], groups,
                                    ("up", i, j, 0), st)
            if i < n - 1:
                p = f"up_blocks.{i}.upsamplers.0.conv"
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")               # [ext] Upsample2D
                x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
    except _Stop:
        return None
    x = F.silu(F.group_norm(x, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
    return F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


# =================================================================================================
# scheduler [ext] DDIMScheduler (eta = 0) — used at pipelines.py:150,196,221,357,443,545,583
# =================================================================================================
class DDIM:
    def __init__(self, prediction_type="epsilon", num_train_timesteps=1000, beta_start=0.00085,
                 beta_end=0.012, steps_offset=1):
        self.T = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]          # set_alpha_to_one=False
        self.steps_offset = steps_offset
        self.prediction_type = prediction_type
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.T // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)) + self.steps_offset

    # ---- utils/schedule.py ------------------------------------------------------------------
    def use_fast_schedule(self, fast_after_steps, fast_rate=2):
        """schedule.py:4-8 as applied at pipelines.py:151-152,358-359: keep the first `fast_after_steps`
        timesteps, then every `fast_rate`-th one."""
        ts = self.timesteps
        if fast_after_steps < len(ts) - 1:
            self.timesteps = torch.cat((ts[:fast_after_steps], ts[fast_after_steps + 1::fast_rate]), dim=0)

    def adjust_inference_steps(self, index, t):
        """schedule.py:10-12 (`dynamically_adjust_inference_steps`, pipelines.py:217-218,439-440): make the
        next `step()` land on the following timestep of an irregular schedule."""
        nxt = int(self.timesteps[index + 1]) if index + 1 < len(self.timesteps) else -1
        self.num_inference_steps = self.T // (int(t) - nxt)

    def step(self, eps, t, x):
        t = int(t)
        prev_t = t - self.T // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        if self.prediction_type == "epsilon":
            x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
            e = eps
        else:  # v_prediction
            x0 = a_t ** 0.5 * x - (1 - a_t) ** 0.5 * eps
            e = a_t ** 0.5 * eps + (1 - a_t) ** 0.5 * x
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * e


# =================================================================================================
# energy (utils/guidance.py)
# =================================================================================================
def scale_proportion(obj_box, H, W):
    """utils/utils.py:57-70 (non-legacy branch; Python round = banker's rounding)."""
    x_min, y_min = round(obj_box[0] * W), round(obj_box[1] * H)
    box_w, box_h = round((obj_box[2] - obj_box[0]) * W), round((obj_box[3] - obj_box[1]) * H)
    x_max, y_max = x_min + box_w, y_min + box_h
    return max(x_min, 0), max(y_min, 0), min(x_max, W), min(y_max, H)


def box_mask(obj_boxes, H, W):
    """mask construction of guidance.py:104-114 / :204-207."""
    mask = torch.zeros(H, W)
    if not isinstance(obj_boxes[0], Iterable):
        obj_boxes = [obj_boxes]
    for b in obj_boxes:
        x0, y0, x1, y1 = scale_proportion(b, H, W)
        mask[y0:y1, x0:x1] = 1
    return mask


def ca_loss_per_map(loss, attn_map, bboxes, object_positions, use_ratio_based_loss=True, fg_top_p=0.2,
                    bg_top_p=0.2, fg_weight=1.0, bg_weight=1.0):
    """add_ca_loss_per_attn_map_to_loss (guidance.py:91-148), both branches.  `use_ratio_based_loss` defaults to True
    exactly as the reference's signature does (:91): LMD / LMD+ pass False explicitly (lmd_plus.py:315,487,
    lmd.py:349,521), the `backward_guidance` plugin passes nothing (backward_guidance.py:99-112) and so runs the
    ratio branch (:118-130): per token and head r = sum(A*M)/sum(A), term mean_heads((1-r)^2)."""
    b, i, _ = attn_map.shape
    H = W = int(math.sqrt(i))
    for obj_idx in range(len(bboxes)):
        obj_loss = 0
        mask = box_mask(bboxes[obj_idx], H, W)
        for pos in object_positions[obj_idx]:
            ca = attn_map[:, :, pos]
            m1 = mask.view(1, -1)
            if use_ratio_based_loss:
                activation = (ca * m1).sum(dim=-1) / ca.sum(dim=-1)
                obj_loss += torch.mean((1 - activation) ** 2)
            else:
                k_fg = (mask.sum() * fg_top_p).long().clamp_(min=1)
                k_bg = ((1 - mask).sum() * bg_top_p).long().clamp_(min=1)
                obj_loss += (1 - (ca * m1).topk(k=k_fg).values.mean(dim=1)).sum(dim=0) * fg_weight
                obj_loss += ((ca * (1 - m1)).topk(k=k_bg).values.mean(dim=1)).sum(dim=0) * bg_weight
        loss += obj_loss / len(object_positions[obj_idx])
    return loss


def ref_ca_loss(loss, saved_attn, bboxes, object_positions, guidance_attn_keys, ref_ca_saved_attns,
                ref_ca_last_token_only, ref_ca_word_token_only, word_token_indices, index, loss_weight,
                eps=1e-5):
    """add_ref_ca_loss_per_attn_map_to_lossv2 (guidance.py:150-242)."""
    if loss_weight == 0.:
        return loss
    for obj_idx in range(len(bboxes)):
        obj_loss = 0
        obj_boxes = bboxes[obj_idx]
        obj_refs = ref_ca_saved_attns[obj_idx]
        if not isinstance(obj_boxes[0], Iterable):
            obj_boxes, obj_refs = [obj_boxes], [obj_refs]
        for obj_box, obj_ref in zip(obj_boxes, obj_refs):
            obj_ref = obj_ref[index]
            for key in guidance_attn_keys:
                attn_map = saved_attn[key].squeeze(dim=0)
                ref_map = obj_ref[key][0, :, :, 0]
                b, i, _ = attn_map.shape
                H = W = int(math.sqrt(i))
                m = box_mask(obj_box, H, W).reshape(1, -1)
                if ref_ca_word_token_only:
                    toks = [word_token_indices[obj_idx]]
                elif ref_ca_last_token_only:
                    toks = [object_positions[obj_idx][-1]]
                else:
                    toks = object_positions[obj_idx]
                for pos in toks:
                    a = attn_map[:, :, pos] * m
                    a = a / (a.sum(dim=-1, keepdim=True) + eps)
                    r = ref_map * m
                    r = r / (r.sum(dim=-1, keepdim=True) + eps)
                    obj_loss += torch.mean((a - r).abs().sum(dim=-1), dim=0)
        loss += loss_weight * obj_loss / (len(obj_boxes) * len(toks))
    return loss


def compute_ca_lossv3(saved_attn, bboxes, object_positions, guidance_attn_keys, ref_ca_saved_attns=None,
                      ref_ca_last_token_only=True, ref_ca_word_token_only=False, word_token_indices=None,
                      index=None, ref_ca_loss_weight=1.0, **kw):
    """guidance.py:244-286."""
    loss = torch.tensor(0.)
    n_obj = len(bboxes)
    if n_obj == 0:
        return loss
    kw = {k: v for k, v in kw.items() if k in ("use_ratio_based_loss", "fg_top_p", "bg_top_p", "fg_weight", "bg_weight")}
    for key in guidance_attn_keys:
        loss = ca_loss_per_map(loss, saved_attn[key].squeeze(dim=0), bboxes, object_positions, **kw)
    n_attn = len(guidance_attn_keys)
    if n_attn > 0:
        loss = loss / (n_obj * n_attn)
    if ref_ca_saved_attns is not None:
        ref = ref_ca_loss(torch.tensor(0.), saved_attn, bboxes, object_positions, guidance_attn_keys,
                          ref_ca_saved_attns, ref_ca_last_token_only, ref_ca_word_token_only,
                          word_token_indices, index, ref_ca_loss_weight)
        loss = loss + ref / (n_obj * n_attn)
    return loss


# =================================================================================================
# samplers (models/pipelines.py)
# =================================================================================================
def latent_backward_guidance(sd, cfg, sched, cond_emb, index, bboxes, object_positions, t, latents, loss,
                             loss_scale=30, loss_threshold=0.2, max_iter=5, max_index_step=10,
                             guidance_attn_keys=None, gligen=None, fuser_enabled=True,
                             early_exit=True, trace=None, **kw):
    """pipelines.py:16-82 (DDIM branch :62-69: step scaled by sqrt(1-alpha_bar_t))."""
    it = 0
    if index < max_index_step:
        if isinstance(max_iter, list):
            max_iter = max_iter[index] if len(max_iter) > index else max_iter[-1]
        while loss.item() / loss_scale > loss_threshold and it < max_iter and index < max_index_step:
            saved = {}
            latents = latents.detach().requires_grad_(True)
            stop = guidance_attn_keys[-1] if early_exit else None
            unet_forward(sd, cfg, latents, t, cond_emb, saved=saved, save_keys=guidance_attn_keys,
                         gligen=gligen, fuser_enabled=fuser_enabled, stop_after=stop)
            loss = compute_ca_lossv3(saved, bboxes, object_positions, guidance_attn_keys, index=index, **kw) * loss_scale
            grad = torch.autograd.grad(loss.requires_grad_(True), [latents])[0]
            latents = latents.detach()
            a_t = sched.alphas_cumprod[int(t)]
            latents = latents - (1 - a_t) ** 0.5 * grad
            loss = loss.detach()
            it += 1
            if trace is not None:
                trace.append(dict(index=index, it=it, loss=float(loss), grad=grad.clone()))
    return latents, loss


def prepare_gligen_condition(bboxes, phrase_embeddings_list, positive_len=768):
    """pipelines.py:285-321 with the CLIP pooler_output replaced by given (n,768) embeddings.
    bboxes: list (batch) of list of boxes."""
    bs, max_objs = len(bboxes), 30
    n_objs = min(max(len(b) for b in bboxes), max_objs)
    boxes = torch.zeros(bs, max_objs, 4)
    emb = torch.zeros(bs, max_objs, positive_len)
    masks = torch.zeros(bs, max_objs)
    if n_objs > 0:
        for i, (bb, pe) in enumerate(zip(bboxes, phrase_embeddings_list)):
            bb = torch.tensor(bb[:n_objs])
            boxes[i, :bb.shape[0]] = bb
            emb[i, :bb.shape[0]] = pe[:bb.shape[0]]
            masks[i, :bb.shape[0]] = 1
    boxes, emb, masks = boxes.repeat(2, 1, 1), emb.repeat(2, 1, 1), masks.repeat(2, 1)
    cond_len = bs * 2
    masks[:cond_len // 2] = 0                                   # :317
    return boxes, emb, masks, cond_len


def _cfg_step(sd, cfg, sched, latents, t, text_emb, gs, **uk):
    with torch.no_grad():
        eps = unet_forward(sd, cfg, torch.cat([latents] * 2), t, text_emb, **uk)
        eu, ec = eps.chunk(2)
        return sched.step(eu + gs * (ec - eu), t, latents)


def generate_semantic_guidance(sd, cfg, sched, latents, input_embeddings, steps, bboxes, object_positions,
                               guidance_scale=7.5, semantic_guidance_kwargs=None, saved_cross_attn_keys=None,
                               return_cond_ca_only=False, return_token_ca_only=None, trace=None):
    """pipelines.py:129-247 -> (latents, saved_attns per step, latents_all)."""
    text_emb, _, cond_emb = input_embeddings
    latents = latents.clone()
    latents_all = [latents]
    sched.set_timesteps(steps)
    loss = torch.tensor(10000.)
    saved_attns = []
    for index, t in enumerate(sched.timesteps):
        if bboxes:
            latents, loss = latent_backward_guidance(sd, cfg, sched, cond_emb, index, bboxes, object_positions,
                                                     t, latents, loss, trace=trace, **semantic_guidance_kwargs)
        saved = {}
        latents = _cfg_step(sd, cfg, sched, latents, t, text_emb, guidance_scale, saved=saved,
                            save_keys=saved_cross_attn_keys, cond_only=return_cond_ca_only,
                            token_only=return_token_ca_only)
        saved_attns.append(saved)
        latents_all.append(latents)
    return latents, saved_attns, torch.stack(latents_all, dim=0)


def generate_partial_frozen(sd, cfg, sched, latents_all, frozen_mask, input_embeddings, steps, frozen_steps,
                            guidance_scale=7.5, bboxes=None, object_positions=None,
                            semantic_guidance_kwargs=None, trace=None, per_step=None):
    """pipelines.py:541-599 (without the inline VAE decode)."""
    text_emb, _, cond_emb = input_embeddings
    sched.set_timesteps(steps)
    frozen_mask = frozen_mask.to(torch.float32).clamp(0., 1.)
    latents = latents_all[0]
    loss = torch.tensor(10000.)
    for index, t in enumerate(sched.timesteps):
        if bboxes:
            latents, loss = latent_backward_guidance(sd, cfg, sched, cond_emb, index, bboxes, object_positions,
                                                     t, latents, loss, trace=trace, **semantic_guidance_kwargs)
        latents = _cfg_step(sd, cfg, sched, latents, t, text_emb, guidance_scale)
        if index < frozen_steps:
            latents = latents_all[index + 1] * frozen_mask + latents * (1. - frozen_mask)
        if per_step is not None:
            per_step.append(latents.clone())
    return latents


def generate_gligen(sd, cfg, sched, latents, input_embeddings, steps, bboxes, phrase_embeddings,
                    gligen_scheduled_sampling_beta=0.3, guidance_scale=7.5, frozen_steps=20, frozen_mask=None,
                    saved_cross_attn_keys=None, return_saved_cross_attn=False, return_cond_ca_only=False,
                    return_token_ca_only=None, semantic_guidance=False, semantic_guidance_bboxes=None,
                    semantic_guidance_object_positions=None, semantic_guidance_kwargs=None, trace=None,
                    per_step=None, dynamic_num_inference_steps=False, fast_after_steps=None, fast_rate=2):
    """pipelines.py:323-473.  bboxes: list of boxes of ONE image; phrase_embeddings (n,768)."""
    text_emb, _, cond_emb = input_embeddings
    latents_all_input = None
    if latents.dim() == 5:
        latents_all_input, latents = latents, latents[0]
    latents = latents.clone()
    latents_all = [latents]
    sched.set_timesteps(steps)
    if fast_after_steps is not None:                                  # :358-359
        sched.use_fast_schedule(fast_after_steps, fast_rate)
    if frozen_mask is not None:
        frozen_mask = frozen_mask.to(torch.float32).clamp(0., 1.)
    boxes, emb, masks, cond_len = prepare_gligen_condition([bboxes], [phrase_embeddings],
                                                           cfg.get("gligen_positive_len", 768))
    guide = bool(semantic_guidance_bboxes) and semantic_guidance
    loss = torch.tensor(10000.)
    g_gligen = dict(boxes=boxes[:cond_len // 2], positive_embeddings=emb[:cond_len // 2],
                    masks=masks[:cond_len // 2])                # :381-384 — the zero-masked half
    m_gligen = dict(boxes=boxes, positive_embeddings=emb, masks=masks)
    n_ground = int(gligen_scheduled_sampling_beta * len(sched.timesteps))
    saved_attns = []
    for index, t in enumerate(sched.timesteps):
        fuser_on = index < n_ground                               # :408-414
        if guide:
            latents, loss = latent_backward_guidance(sd, cfg, sched, cond_emb, index, semantic_guidance_bboxes,
                                                     semantic_guidance_object_positions, t, latents, loss,
                                                     gligen=g_gligen, fuser_enabled=fuser_on, trace=trace,
                                                     **semantic_guidance_kwargs)
        saved = {} if return_saved_cross_attn else None
        if dynamic_num_inference_steps:                               # :439-440
            sched.adjust_inference_steps(index, t)
        latents = _cfg_step(sd, cfg, sched, latents, t, text_emb, guidance_scale, saved=saved,
                            save_keys=saved_cross_attn_keys, cond_only=return_cond_ca_only,
                            token_only=return_token_ca_only, gligen=m_gligen, fuser_enabled=fuser_on)
        if frozen_mask is not None and index < frozen_steps:
            latents = latents_all_input[index + 1] * frozen_mask + latents * (1. - frozen_mask)
        saved_attns.append(saved)
        if fast_after_steps is None or index < fast_after_steps:      # :449 "do not save the latents in the fast steps"
            latents_all.append(latents)
        if per_step is not None:
            per_step.append(latents.clone())
    return latents, saved_attns, torch.stack(latents_all, dim=0)


# =================================================================================================
# host-side latent preparation (utils/latents.py, utils/utils.py)
# =================================================================================================
def get_unscaled_latents(seed, in_channels, h, w):
    """latents.py:7-18: CPU generator, fp32."""
    return torch.randn((1, in_channels, h, w), generator=torch.manual_seed(seed), dtype=torch.float32)


def proportion_to_mask(box, H, W):
    x0, y0, x1, y1 = scale_proportion(box, H, W)
    m = torch.zeros(H, W)
    m[y0:y1, x0:x1] = 1.
    return m


def get_input_latents_list(bg_seed, fg_seed_start, so_boxes, fg_blending_ratio, in_channels=4, H=64, W=64):
    """latents.py:120-161 (+ blend_latents :25-36)."""
    bg = get_unscaled_latents(bg_seed, in_channels, H, W)
    out = []
    for idx, box in enumerate(so_boxes):
        m = proportion_to_mask(box, H, W)
        fg_seed = fg_seed_start + idx
        if fg_seed == bg_seed:
            fg_seed += 12345
        fg = get_unscaled_latents(fg_seed, in_channels, H, W)
        out.append(bg * (1. - m) + (bg * np.sqrt(1. - fg_blending_ratio) + fg * np.sqrt(fg_blending_ratio)) * m)
    return out, bg


def binary_mask_to_box_mask(mask):
    """utils/utils.py:72-100."""
    loc = torch.where(mask)
    h, w = mask.shape
    ymin, ymax = max(min(loc[0]) - 1, 0), min(max(loc[0]) + 1, h)
    xmin, xmax = max(min(loc[1]) - 1, 0), min(max(loc[1]) + 1, w)
    m = torch.zeros(h, w)
    m[ymin:ymax + 1, xmin:xmax + 1] = 1.
    return m


def compose_latents(latents_all_list, mask_tensor_list, steps, latents_bg):
    """latents.py:38-83 (compose_box_to_bg=True, no fast schedule)."""
    composed = torch.zeros((steps + 1, *latents_bg.shape))
    composed[0] = latents_bg
    fg_idx = torch.zeros(latents_bg.shape[-2:], dtype=torch.long)
    order = np.argsort(-np.array([m.sum().item() for m in mask_tensor_list]))
    for i in order:
        bm = binary_mask_to_box_mask(mask_tensor_list[i])[None, None, None]
        composed[0] = composed[0] * (1. - bm) + latents_all_list[i][0] * bm
    for i in order:
        m = mask_tensor_list[i]
        fg_idx = fg_idx * (~m) + (i + 1) * m
        me = m[None, None, None].float()
        composed = composed * (1. - me) + latents_all_list[i] * me
    return composed, fg_idx
