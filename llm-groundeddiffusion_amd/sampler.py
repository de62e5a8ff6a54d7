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
OBJ_KEY_DEFAULT = ("down", 2, 1, 0)                                                                  # lmd_plus.py:380


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


class HipGraph:
    """A captured hipGraph of a launch sequence (torch.cuda.CUDAGraph drives hipStreamBeginCapture on
    torch's current stream — the stream every lgd_* call is enqueued on).  Replaces ~400 host-side
    launches per UNet call by one hipGraphLaunch; everything that varies between replays (timestep,
    frozen-step count, latents, maps) lives in device memory at fixed addresses."""

    def __init__(self, fn, warmup: int = 1):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()                       # also triggers one-time hipFuncSetAttribute calls
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn()

    def __call__(self):
        self.graph.replay()


class _State:
    """Persistent device buffers of one (latent shape, step count): fixed addresses for the graphs."""

    def __init__(self, dev, C, L, T):
        self.lat = torch.zeros((1, C, L, L), device=dev, dtype=F32)
        self.hist = torch.zeros((T + 1, 1, C, L, L), device=dev, dtype=F32)
        self.frozen_ref = torch.zeros((T + 1, 1, C, L, L), device=dev, dtype=F32)
        self.mask = torch.zeros((1, L * L), device=dev, dtype=F32)
        self.ctab = torch.zeros((T, 4), device=dev, dtype=F32)
        self.gtab = torch.zeros((T, 4), device=dev, dtype=F32)
        self.graphs = {}


class LMDSampler:
    def __init__(self, engine: UNetEngine, scheduler: Optional[DDIMScheduler] = None, vae=None,
                 grad_scale: float = 1024.0, use_graphs: bool = True):
        self.eng = engine
        self.dev = engine.device
        self.scheduler = scheduler or DDIMScheduler(prediction_type=engine.cfg.prediction_type)
        self.vae = vae
        self.grad_scale = grad_scale
        self.use_graphs = use_graphs
        self.stats = dict(unet_main=0, guidance_iters=0)
        self._states = {}

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
    def _state(self, C, L, T) -> _State:
        key = (C, L, T)
        if key not in self._states:
            self._states[key] = _State(self.dev, C, L, T)
        return self._states[key]

    def _runner(self, st: _State, name, fn):
        """fn enqueued eagerly or as a cached hipGraph."""
        if not self.use_graphs:
            return fn
        if name not in st.graphs:
            st.graphs[name] = HipGraph(fn)
        return st.graphs[name]

    def backward_guidance(self, gs: GuidanceState, plan_g, index: int, st: _State, fwd_run, bwd_run,
                          trace: Optional[list] = None):
        """latent_backward_guidance (pipelines.py:16-82) on the persistent latent buffer st.lat."""
        if gs is None or index >= gs.max_index_step:
            return
        en = gs.energy
        max_it = gs.iters_at(index)
        it = 0
        en.bind(plan_g.maps, plan_g.gmaps)
        while it < max_it and gs.current_loss() / gs.loss_scale > gs.loss_threshold:
            fwd_run()                                               # grad-plan forward from st.lat
            gs.loss_dev = en.run(self.eng.dyn, grad_scale=self.grad_scale)
            bwd_run()                                               # backward + latent update
            if trace is not None:
                trace.append(dict(index=index, it=it, loss=float(gs.loss_dev.item()),
                                  grad=plan_g.g_latents.clone()))
            it += 1
            gs.iterations += 1
            self.stats["guidance_iters"] += 1

    # ------------------------------------------------------------------------------------------
    def profile_passes(self, L: int, T: int, gligen: bool, guidance_keys=None):
        """Eager (non-graph) launch sequences of the plans the sampler replays, for per-kernel HIP-event
        timing by bench.py: [(name, callable)].  Uses whatever run constants are currently loaded."""
        eng = self.eng
        st = self._state(eng.cfg.in_channels, L, T)
        keys = [tuple(k) for k in (guidance_keys or DEFAULT_GUIDANCE_ATTN_KEYS)]
        plan_keys = sorted({OBJ_KEY_DEFAULT, *DEFAULT_GUIDANCE_ATTN_KEYS})
        out = []
        for f in ([True, False] if gligen else [False]):
            plan = eng.plan(2, L, fuser=f, save_keys=plan_keys)
            out.append((f"main_fuser_{'on' if f else 'off'}", plan.forward))
            pg = eng.plan(1, L, grad=True, fuser=f, stop_key=keys[-1], save_keys=keys, text_batch_offset=1)

            def guide(pg=pg):
                pg.forward(st.lat)
                pg.backward(self.grad_scale)
            out.append((f"guide_fuser_{'on' if f else 'off'}", guide))
        return out

    # ------------------------------------------------------------------------------------------
    def guidance_only(self, latents: torch.Tensor, cond_embeddings: torch.Tensor, num_inference_steps: int,
                      index: int, guidance: dict, *, gligen=None, fuser: bool = False,
                      trace: Optional[list] = None):
        """One latent_backward_guidance call (pipelines.py:16-82) at step `index` of a T-step schedule.
        Returns (latents, loss) like the reference."""
        eng, sch, dev = self.eng, self.scheduler, self.dev
        _, C, L, _ = latents.shape
        T = num_inference_steps
        st = self._state(C, L, T)
        sch.set_timesteps(T)
        st.gtab.copy_(sch.guidance_step_table(dev))
        g = dict(guidance)
        gkeys = [tuple(k) for k in (g.get("guidance_attn_keys") or DEFAULT_GUIDANCE_ATTN_KEYS)]
        gs = g.pop("state", None) or self.make_guidance(L, g.pop("bboxes"), g.pop("object_positions"), **g)
        pg = eng.plan(1, L, grad=True, fuser=fuser, stop_key=gkeys[-1], save_keys=gkeys,
                      text_batch_offset=1, obj_batch_offset=0)
        eng.prepare_timesteps([int(t) for t in sch.timesteps])
        cond = cond_embeddings.to(dev)
        eng.prepare_text(torch.cat([torch.zeros_like(cond), cond]))
        if gligen is not None:
            eng.prepare_gligen(boxes=gligen[0], positive_embeddings=gligen[1], masks=gligen[2])
        eng.set_step(index)

        def g_fwd():
            pg.forward(st.lat)

        def g_bwd():
            ops.axpy(pg.backward(self.grad_scale), st.lat, st.gtab, eng.dyn, 0)
        name = ("guide", fuser, tuple(gkeys))
        gf, gb = self._runner(st, name + ("fwd",), g_fwd), self._runner(st, name + ("bwd",), g_bwd)
        st.lat.copy_(latents.to(dev, F32))
        if gs is not None:
            self.backward_guidance(gs, pg, index, st, gf, gb, trace)
        return st.lat.clone(), (gs.current_loss() if gs is not None else None), gs

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
        latents_all_input = latents if latents.dim() == 5 else None
        lat0 = latents[0] if latents_all_input is not None else latents
        B1, C, L, _ = lat0.shape
        assert B1 == 1
        T = num_inference_steps
        st = self._state(C, L, T)
        sch.set_timesteps(T)
        st.ctab.copy_(sch.coef_table(guidance_scale, dev))
        st.gtab.copy_(sch.guidance_step_table(dev))
        use_gligen = gligen is not None
        n_ground = int(gligen_scheduled_sampling_beta * T) if use_gligen else 0
        save_keys = [tuple(k) for k in saved_cross_attn_keys]
        # a superset of keys is always captured by the main plans so that one graph serves every caller
        plan_keys = sorted(set(save_keys) | {OBJ_KEY_DEFAULT, *DEFAULT_GUIDANCE_ATTN_KEYS}) if save_keys else []

        def fuser_at(index):
            return bool(use_gligen and index < n_ground)                     # pipelines.py:408-414
        gs = None
        gkeys = None
        if guidance is not None:
            g = dict(guidance)
            gkeys = [tuple(k) for k in (g.get("guidance_attn_keys") or DEFAULT_GUIDANCE_ATTN_KEYS)]
            gs = self.make_guidance(L, g.pop("bboxes"), g.pop("object_positions"), **g)

        # ---- plans + launch sequences (built/captured once per shape, cached)
        main_run, guide_run, plans_main, plans_guide = {}, {}, {}, {}
        for f in {fuser_at(i) for i in range(T)}:
            plan = plans_main[f] = eng.plan(2, L, fuser=f, save_keys=plan_keys)

            def main_fn(plan=plan):
                plan.latents_in.copy_(st.lat.expand(2, C, L, L))             # torch.cat([latents]*2)
                plan.forward()
                ops.cfg_ddim_step(plan.eps_out, st.lat, st.lat, st.ctab, eng.dyn, frozen_ref=st.frozen_ref,
                                  mask=st.mask, hist=st.hist)
            main_run[f] = (main_fn, ("main", f, tuple(plan_keys)))
        if gs is not None:
            for f in {fuser_at(i) for i in range(min(gs.max_index_step, T))}:
                pg = plans_guide[f] = eng.plan(1, L, grad=True, fuser=f, stop_key=gkeys[-1], save_keys=gkeys,
                                               text_batch_offset=1, obj_batch_offset=0)

                def g_fwd(pg=pg):
                    pg.forward(st.lat)

                def g_bwd(pg=pg):
                    grad = pg.backward(self.grad_scale)
                    ops.axpy(grad, st.lat, st.gtab, eng.dyn, 0)              # pipelines.py:62-69
                guide_run[f] = (g_fwd, g_bwd, ("guide", f, tuple(gkeys)))

        # ---- per-run constants (before graph capture so that warm-up launches see valid inputs)
        eng.prepare_timesteps([int(t) for t in sch.timesteps])
        eng.prepare_text(text_embeddings)
        if use_gligen:
            eng.prepare_gligen(boxes=gligen[0], positive_embeddings=gligen[1], masks=gligen[2])
        eng.set_step(0)
        eng.dyn[1:2].fill_(0)
        runners_main = {f: self._runner(st, name, fn) for f, (fn, name) in main_run.items()}
        runners_guide = {f: (self._runner(st, name + ("fwd",), gf), self._runner(st, name + ("bwd",), gb))
                         for f, (gf, gb, name) in guide_run.items()}

        # ---- state of this call
        st.lat.copy_(lat0.to(dev, F32))
        st.hist[0].copy_(st.lat)
        if frozen_mask is not None and frozen_steps > 0 and latents_all_input is not None:
            st.frozen_ref.copy_(latents_all_input.to(dev, F32))
            st.mask.copy_(frozen_mask.to(dev, F32).clamp(0., 1.).reshape(1, L * L))
            eng.dyn[1:2].fill_(int(frozen_steps))
        saved = {}
        hw = self.map_hw(L)
        tok = return_token_ca_only
        for k in save_keys:
            Tp = 1 if tok is not None else eng.text_len
            Bp = 1 if return_cond_ca_only else 2
            saved[k] = torch.zeros((T, Bp, self.heads_of(k), hw[k], Tp), device=dev, dtype=F32)

        for index in range(T):
            eng.set_step(index)
            fuser_on = fuser_at(index)
            if gs is not None and index < gs.max_index_step:
                gf, gb = runners_guide[fuser_on]
                self.backward_guidance(gs, plans_guide[fuser_on], index, st, gf, gb, trace)
            runners_main[fuser_on]()
            self.stats["unet_main"] += 1
            if save_keys:
                maps = plans_main[fuser_on].maps
                for k in save_keys:                                   # attention_processor.py:466-476
                    m = maps[k][1:] if return_cond_ca_only else maps[k]
                    saved[k][index].copy_(m[..., int(tok):int(tok) + 1] if tok is not None else m)
        return dict(latents=st.lat.clone(), latents_all=st.hist.clone() if save_all_latents else None,
                    saved=saved, guidance_iters=gs.iterations if gs is not None else 0)

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
