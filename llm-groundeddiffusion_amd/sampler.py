"""Stage-2 sampler loops of LMD / LMD+ on the HIP engine.

One generic denoising loop covers the three sampler functions of the reference
(models/pipelines.py):  generate_gligen :323-473 (LMD+ per-box and overall stages),
generate_semantic_guidance :129-247 (LMD per-box stage / backward_guidance baseline) and
generate_partial_frozen :541-599 (LMD overall stage), plus latent_backward_guidance :16-82.

Per step (all on one HIP stream, no host round trip except the reference's own `loss.item()`):
    [guidance]  while loss/scale > thr and it < max_iter[index]:      (pipelines.py:30)
                    grad-plan forward (B=1, cond only, stops at the last guidance key)
                    energy kernel  -> loss, d loss / d maps
                    grad-plan backward -> d loss / d latents ; latents -= sqrt(1-abar_t) * grad
    main plan forward (B=2: [uncond; cond], optional cross-attention map capture)
    fused CFG + DDIM + frozen-mask blend + history write
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .energy import EnergyTables
from .scheduler import DDIMScheduler
from .unet import N_OBJ_TOKENS, UNetEngine

F32 = torch.float32
DEFAULT_GUIDANCE_ATTN_KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]  # pipelines.py:14


def prepare_gligen_condition(bboxes, phrase_embeddings, device, positive_len=768):
    """pipelines.py:285-321 for ONE image, with the CLIP pooler_output of the phrases given
    (cached layouts: text encoding is outside the hot path).  Returns boxes (2,30,4), embeddings
    (2,30,768), masks (2,30) with the unconditional half zero-masked (:317)."""
    n = min(len(bboxes), N_OBJ_TOKENS)
    boxes = torch.zeros((1, N_OBJ_TOKENS, 4), dtype=F32)
    emb = torch.zeros((1, N_OBJ_TOKENS, positive_len), dtype=F32)
    masks = torch.zeros((1, N_OBJ_TOKENS), dtype=F32)
    if n > 0:
        boxes[0, :n] = torch.tensor(bboxes[:n], dtype=F32)
        emb[0, :n] = torch.as_tensor(phrase_embeddings, dtype=F32).cpu()[:n]
        masks[0, :n] = 1
    boxes, emb, masks = boxes.repeat(2, 1, 1), emb.repeat(2, 1, 1), masks.repeat(2, 1)
    masks[:1] = 0
    return boxes.to(device), emb.to(device), masks.to(device)


class GuidanceState:
    """Everything the guidance inner loop needs for one layout."""

    def __init__(self, energy: EnergyTables, loss_scale, loss_threshold, max_iter, max_index_step):
        self.energy = energy
        self.loss_scale, self.loss_threshold = float(loss_scale), float(loss_threshold)
        self.max_iter, self.max_index_step = max_iter, int(max_index_step)
        self.loss = 10000.0             # pipelines.py:161,375,552
        self.loss_dev = None            # device scalar of the last launched energy, read lazily
        self.iterations = 0

    def current_loss(self) -> float:
        if self.loss_dev is not None:
            self.loss = float(self.loss_dev.item())     # host sync, as pipelines.py:30
            self.loss_dev = None
        return self.loss

    def iters_at(self, index):
        m = self.max_iter
        if isinstance(m, (list, tuple)):
            m = m[index] if len(m) > index else m[-1]
        return int(m)


class LMDSampler:
    def __init__(self, engine: UNetEngine, scheduler: Optional[DDIMScheduler] = None, vae=None,
                 grad_scale: float = 1024.0):
        self.eng = engine
        self.dev = engine.device
        self.scheduler = scheduler or DDIMScheduler(prediction_type=engine.cfg.prediction_type)
        self.vae = vae
        self.grad_scale = grad_scale
        self.stats = dict(unet_main=0, guidance_iters=0)

    # ------------------------------------------------------------------------------------------
    def map_hw(self, L: int) -> Dict[Tuple, int]:
        out = {}
        for b in self.eng.blocks:
            for a in b.attns:
                side = L >> b.level
                out[a.key] = side * side
        return out

    def heads_of(self, key) -> int:
        for b in self.eng.blocks:
            for a in b.attns:
                if a.key == tuple(key):
                    return a.heads
        raise KeyError(key)

    def make_guidance(self, L, bboxes, object_positions, *, loss_scale=30, loss_threshold=0.2, max_iter=5,
                      max_index_step=10, guidance_attn_keys=None, ref_maps=None, **kw) -> Optional[GuidanceState]:
        """kwargs as latent_backward_guidance / compute_ca_lossv3 receive them (pipelines.py:16,
        guidance.py:244).  ref_maps: fp32 [T][n_boxes_flat][n_keys][heads][max_hw] or None."""
        if not bboxes or max_index_step <= 0:
            return None
        keys = [tuple(k) for k in (guidance_attn_keys or DEFAULT_GUIDANCE_ATTN_KEYS)]
        heads = self.heads_of(keys[0])
        assert all(self.heads_of(k) == heads for k in keys), "guidance keys with different head counts"
        ekw = {k: kw[k] for k in ("fg_top_p", "bg_top_p", "fg_weight", "bg_weight", "ref_ca_loss_weight",
                                  "ref_ca_word_token_only", "ref_ca_last_token_only", "word_token_indices")
               if k in kw}
        en = EnergyTables(self.dev, bboxes, object_positions, keys, self.map_hw(L), heads,
                          self.eng.text_len, loss_scale=loss_scale, ref_boxes=ref_maps is not None, **ekw)
        if ref_maps is not None:
            T = ref_maps.shape[0]
            en.set_refs(ref_maps.reshape(T, -1, heads, en.max_hw))
        return GuidanceState(en, loss_scale, loss_threshold, max_iter, max_index_step)

    # ------------------------------------------------------------------------------------------
    def backward_guidance(self, gs: GuidanceState, plan_g, index: int, latents: torch.Tensor,
                          gtable: torch.Tensor, trace: Optional[list] = None):
        """latent_backward_guidance (pipelines.py:16-82).  `latents` (1,C,L,L) fp32 is updated in place."""
        if gs is None or index >= gs.max_index_step:
            return
        en = gs.energy
        max_it = gs.iters_at(index)
        it = 0
        en.bind(plan_g.maps, plan_g.gmaps)
        while it < max_it and gs.current_loss() / gs.loss_scale > gs.loss_threshold:
            plan_g.forward(latents)
            gs.loss_dev = en.run(index, grad_scale=self.grad_scale)
            grad = plan_g.backward(self.grad_scale)
            if trace is not None:
                trace.append(dict(index=index, it=it, loss=float(gs.loss_dev.item()), grad=grad.clone()))
            ops.axpy(grad, latents, gtable, self.eng.step_idx, 0)
            it += 1
            gs.iterations += 1
            self.stats["guidance_iters"] += 1

    # ------------------------------------------------------------------------------------------
    def denoise(self, latents: torch.Tensor, text_embeddings: torch.Tensor, num_inference_steps: int, *,
                guidance_scale: float = 7.5,
                gligen: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
                gligen_scheduled_sampling_beta: float = 0.3,
                guidance: Optional[dict] = None,
                frozen_steps: int = 0, frozen_mask: Optional[torch.Tensor] = None,
                saved_cross_attn_keys: Sequence[Tuple] = (), return_cond_ca_only: bool = False,
                return_token_ca_only: Optional[int] = None, save_all_latents: bool = True,
                trace: Optional[list] = None):
        """Generic 50-step loop.

        latents: (1,C,L,L) start latents or (T+1,1,C,L,L) history whose [0] is the start and whose
          [i+1] feeds the frozen-mask blend (pipelines.py:340-345, 445-446, 585-586).
        text_embeddings: (2,77,Cx) = [uncond; cond].
        gligen: (boxes (2,30,4), embeddings (2,30,768), masks (2,30)) or None.
        guidance: dict(bboxes, object_positions, **semantic_guidance_kwargs, [ref_maps]) or None.
        Returns dict(latents, latents_all (T+1,1,C,L,L) device, saved {key: fp32 [T,Bp,H,HW,Tp]},
        guidance_iters).
        """
        eng, sch, dev = self.eng, self.scheduler, self.dev
        latents_all_input = None
        if latents.dim() == 5:
            latents_all_input = latents.to(dev, F32).contiguous()
            latents = latents_all_input[0]
        lat = latents.to(dev, F32).clone().contiguous()
        B1, C, L, _ = lat.shape
        assert B1 == 1
        T = num_inference_steps
        sch.set_timesteps(T)
        ctab = sch.coef_table(guidance_scale, dev)
        gtab = sch.guidance_step_table(dev)
        use_gligen = gligen is not None
        n_ground = int(gligen_scheduled_sampling_beta * T) if use_gligen else 0
        save_keys = [tuple(k) for k in saved_cross_attn_keys]

        # ---- plans (built once per shape, cached on the engine)
        def main_plan(fuser):
            return eng.plan(2, L, fuser=fuser, save_keys=save_keys, save_cond_only=return_cond_ca_only)
        gs = None
        gkeys = None
        if guidance is not None:
            g = dict(guidance)
            gkeys = [tuple(k) for k in (g.get("guidance_attn_keys") or DEFAULT_GUIDANCE_ATTN_KEYS)]
            gs = self.make_guidance(L, g.pop("bboxes"), g.pop("object_positions"), **g)

        def guide_plan(fuser):
            return eng.plan(1, L, grad=True, fuser=fuser, stop_key=gkeys[-1], save_keys=gkeys,
                            text_batch_offset=1, obj_batch_offset=0)
        def fuser_at(index):
            return bool(use_gligen and index < n_ground)                     # pipelines.py:408-414
        plans_main = {f: main_plan(f) for f in {fuser_at(i) for i in range(T)}}
        plans_guide = {}
        if gs is not None:
            plans_guide = {f: guide_plan(f) for f in {fuser_at(i) for i in range(min(gs.max_index_step, T))}}

        # ---- per-run constants
        eng.prepare_timesteps([int(t) for t in sch.timesteps])
        eng.prepare_text(text_embeddings)
        if use_gligen:
            eng.prepare_gligen(boxes=gligen[0], positive_embeddings=gligen[1], masks=gligen[2])

        hist = torch.zeros((T + 1, 1, C, L, L), device=dev, dtype=F32) if save_all_latents else None
        if hist is not None:
            hist[0].copy_(lat)
        saved = {}
        hw = self.map_hw(L)
        for k in save_keys:
            Tp = 1 if return_token_ca_only is not None else eng.text_len
            Bp = 1 if return_cond_ca_only else 2
            saved[k] = torch.zeros((T, Bp, self.heads_of(k), hw[k], Tp), device=dev, dtype=F32)
        mask_dev = None
        if frozen_mask is not None and frozen_steps > 0:
            mask_dev = frozen_mask.to(dev, F32).clamp(0., 1.).reshape(1, L * L).contiguous()
        lat_next = torch.empty_like(lat)
        tok = -1 if return_token_ca_only is None else int(return_token_ca_only)

        for index in range(T):
            eng.set_step(index)
            fuser_on = fuser_at(index)
            if gs is not None and index < gs.max_index_step:
                self.backward_guidance(gs, plans_guide[fuser_on], index, lat, gtab, trace)
            plan = plans_main[fuser_on]
            for k in save_keys:
                plan.map_sink[k][0] = saved[k][index]
                plan.map_sink[k][1] = tok
            plan.latents_in.copy_(lat.expand(2, C, L, L))                    # torch.cat([latents]*2)
            eps = plan.forward()
            self.stats["unet_main"] += 1
            ops.cfg_ddim_step(eps, lat, lat_next, ctab, eng.step_idx,
                              frozen_ref=latents_all_input if mask_dev is not None else None,
                              mask=mask_dev, frozen_steps=frozen_steps, hist=hist)
            lat, lat_next = lat_next, lat
        return dict(latents=lat, latents_all=hist, saved=saved,
                    guidance_iters=gs.iterations if gs is not None else 0)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, latents: torch.Tensor):
        """pipelines.py:117-127: VAE decode -> uint8 HWC (the VAE is [ext] and stays PyTorch/MIOpen)."""
        if self.vae is None:
            raise RuntimeError("no VAE attached to the sampler")
        image = self.vae.decode(latents / 0.18215)
        image = (image / 2 + 0.5).clamp(0, 1)
        image = image.detach().float().cpu().permute(0, 2, 3, 1).numpy()
        return (image * 255).round().astype("uint8")
