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

import threading

import torch

from . import ops
from .energy import BoxDiffTables, EnergyTables
from .lanes import GATE
from .scheduler import DDIMScheduler
from .unet import N_OBJ_TOKENS, UNetEngine

F32 = torch.float32
DEFAULT_GUIDANCE_ATTN_KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]  # pipelines.py:14
OBJ_KEY_DEFAULT = ("down", 2, 1, 0)                                                                  # lmd_plus.py:380
BOXDIFF_GUIDANCE_ATTN_KEYS = [("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]  # generation/boxdiff.py:33-39


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
    """Everything the guidance inner loop needs for one image (layout)."""

    def __init__(self, energy, loss_scale, loss_threshold, max_iter, max_index_step, kind="lmd", step_scale=None):
        self.energy = energy
        self.loss_scale, self.loss_threshold = float(loss_scale), float(loss_threshold)
        self.max_iter, self.max_index_step = max_iter, int(max_index_step)
        # "lmd": latent_backward_guidance (pipelines.py:16-82: loss-thresholded inner loop, step sqrt(1 - abar_t));
        # "boxdiff": utils/boxdiff.py:199-259 (ONE step per denoising step, step_scale(index, n_timesteps) de-scaled by
        # the amp loss scale)
        self.kind, self.step_scale = kind, step_scale
        self.loss = 10000.0             # carried across steps; pipelines.py:161,375,552
        self.iterations = 0
        self.iterations_fuser_on = 0    # of which: taken while the GLIGEN fuser was enabled (accounting only)

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
        # exclusive among the host threads of a lanes.LanePool: other lanes park at their next step boundary
        with GATE.exclusive():
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    fn()                       # also triggers one-time hipFuncSetAttribute calls
            cur.wait_stream(side)
            cur.synchronize()
            side.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                fn()

    def __call__(self):
        self.graph.replay()


class _State:
    """Persistent device buffers of one (batch, latent shape, step count): fixed addresses for the graphs."""

    def __init__(self, dev, nb, C, L, T):
        self.lat = torch.zeros((nb, C, L, L), device=dev, dtype=F32)
        self.hist = torch.zeros((T + 1, nb, C, L, L), device=dev, dtype=F32)
        self.frozen_ref = torch.zeros((T + 1, nb, C, L, L), device=dev, dtype=F32)
        self.mask = torch.zeros((nb, L * L), device=dev, dtype=F32)
        self.active = torch.zeros(nb, device=dev, dtype=F32)      # per-image guidance on/off
        self.ctab = torch.zeros((T, 4), device=dev, dtype=F32)
        self.mtab = torch.zeros((T, 8), device=dev, dtype=F32)      # linear-multistep schedulers (DPM-Solver++)
        self.x0_prev = torch.zeros((nb, C, L, L), device=dev, dtype=F32)
        self.gtab = torch.zeros((T, 4), device=dev, dtype=F32)
        self.graphs = {}


def plan_chunks(n: int, cap: int, buckets: Sequence[int], max_pad: float = 0.25) -> List[Tuple[int, int]]:
    """How `n` independent images are packed into UNet calls: a list of (images, bucket) with bucket >= images, bucket in
    `buckets`, bucket <= cap.  A call is padded up to its bucket with inert copies (so that only a handful of launch
    plans, captured graphs and tuned GEMM shapes exist), but only while the padding stays within `max_pad` of the call;
    otherwise the largest bucket that FITS is split off and the remainder is planned again: 5 -> 4 + 1 (0 padded)
    instead of 8 (3 padded), 11 -> 8 + 3(+1), 7 -> 8.  Pure function: the CPU dry run of bench.py prices a rank's padded
    work with it."""
    bs = sorted(b for b in buckets if b <= max(cap, 1)) or [1]
    out: List[Tuple[int, int]] = []
    left = int(n)
    while left > 0:
        up = next((b for b in bs if b >= left), None)
        if up is not None and (up - left) <= max_pad * up:
            out.append((left, up))
            break
        down = max(b for b in bs if b <= left)
        out.append((down, down))
        left -= down
    return out


class Job:
    """One image of a batched denoising call."""

    def __init__(self, latents, text, gligen=None, guidance=None, frozen_mask=None, token=None):
        self.latents, self.text, self.gligen = latents, text, gligen
        self.guidance, self.frozen_mask, self.token = guidance, frozen_mask, token


class LMDSampler:
    BUCKETS = (1, 2, 4, 8, 16, 32)

    def __init__(self, engine: UNetEngine, scheduler: Optional[DDIMScheduler] = None, vae=None,
                 grad_scale: float = 1024.0, use_graphs: bool = True, max_batch: int = 8,
                 max_batch_guided: int = 4):
        """max_batch / max_batch_guided: images per UNet call for unguided / guided denoising calls; longer
        job lists are chunked (`plan_chunks`), and a chunk is padded to the next size in BUCKETS — by at most
        `max_pad` of the call, else it is split — so that only a handful of launch plans, captured graphs and
        (tuned) GEMM shapes ever exist."""
        self.max_batch = max(1, min(int(max_batch), engine.max_text_batch // 2))
        self.max_batch_guided = max(1, min(int(max_batch_guided), self.max_batch))
        self.eng = engine
        self.dev = engine.device
        self.scheduler = scheduler or DDIMScheduler(prediction_type=engine.cfg.prediction_type)
        self.vae = vae
        self.grad_scale = grad_scale
        self.use_graphs = use_graphs
        self.max_pad = 0.25          # plan_chunks: a call is padded by at most this fraction of its bucket
        self.stats = dict(unet_main=0, guidance_iters=0, images=0, padded_images=0)
        self.pass_counts: Dict[Tuple, int] = {}      # (kind, fuser on?, images) -> launches of that plan
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

    def make_guidance(self, lat_size, bboxes, object_positions, *, loss_scale=30, loss_threshold=0.2, max_iter=5,
                      max_index_step=None, guidance_attn_keys=None, ref_maps=None, **kw) -> Optional[GuidanceState]:
        """kwargs as latent_backward_guidance / compute_ca_lossv3 receive them (pipelines.py:16,
        guidance.py:244) — including the reference's default `use_ratio_based_loss=True` (guidance.py:91) when the
        caller does not switch it off.  ref_maps: fp32 [T][n_boxes_flat][n_keys][heads][max_hw] or None.
        lat_size: side of the latent map (NOT called `L`: BoxDiff's corner-margin hyperparameter has that name in the
        reference, utils/boxdiff.py:28,169, and arrives through **kw).  max_index_step None = the callee's own default:
        10 for latent_backward_guidance (pipelines.py:16), 25 for latent_backward_guidance_boxdiff (utils/boxdiff.py:190)."""
        if kw.get("use_boxdiff"):
            # dispatched BEFORE this function's defaults apply; reference arguments the port cannot honour are errors, not ignored
            if not bboxes or (max_index_step is not None and max_index_step <= 0):
                return None
            bkw = {k: v for k, v in kw.items() if k not in ("use_boxdiff", "verbose")}
            if bkw.pop("normalize_eot", False):
                raise NotImplementedError("normalize_eot=True: the reference asserts against it as well (utils/boxdiff.py)")
            if max_index_step is not None:
                bkw["max_index_step"] = max_index_step
            return self.make_boxdiff_guidance(lat_size, bboxes, object_positions, guidance_attn_keys=guidance_attn_keys, **bkw)
        L = lat_size
        max_index_step = 10 if max_index_step is None else max_index_step
        if not bboxes or max_index_step <= 0:
            return None
        keys = [tuple(k) for k in (guidance_attn_keys or DEFAULT_GUIDANCE_ATTN_KEYS)]
        heads = self.heads_of(keys[0])
        assert all(self.heads_of(k) == heads for k in keys), "guidance keys with different head counts"
        ekw = {k: kw[k] for k in ("use_ratio_based_loss", "fg_top_p", "bg_top_p", "fg_weight", "bg_weight", "ref_ca_loss_weight",
                                  "ref_ca_word_token_only", "ref_ca_last_token_only", "word_token_indices")
               if k in kw}
        en = EnergyTables(self.dev, bboxes, object_positions, keys, self.map_hw(L), heads,
                          self.eng.text_len, loss_scale=loss_scale, ref_boxes=ref_maps is not None, **ekw)
        if ref_maps is not None:
            T = ref_maps.shape[0]
            en.set_refs(ref_maps.reshape(T, -1, heads, en.max_hw))
        return GuidanceState(en, loss_scale, loss_threshold, max_iter, max_index_step)

    def make_boxdiff_guidance(self, lat_size, bboxes, object_positions, *, max_index_step=25, guidance_attn_keys=None,
                              amp_loss_scale=10, latent_scale=20, scale_range=(1., 0.5), **kw) -> GuidanceState:
        """latent_backward_guidance_boxdiff's arguments (utils/boxdiff.py:199): one gradient step per denoising step while
        index < max_index_step — no loss threshold, no inner loop — of size latent_scale * sqrt(ramp(index)), the loss
        scaled by amp_loss_scale and the step de-scaled by it (:224,:236-238)."""
        keys = [tuple(k) for k in (guidance_attn_keys or BOXDIFF_GUIDANCE_ATTN_KEYS)]
        heads = self.heads_of(keys[0])
        assert all(self.heads_of(k) == heads for k in keys), "guidance keys with different head counts"
        known = ("P", "L", "smooth_attentions", "sigma", "kernel_size")
        unknown = sorted(k for k in kw if k not in known and k not in ("loss_scale", "loss_threshold", "max_iter", "ref_maps"))
        if unknown:
            raise TypeError(f"make_boxdiff_guidance: unsupported arguments {unknown}")
        ekw = {k: kw[k] for k in known if k in kw}
        en = BoxDiffTables(self.dev, bboxes, object_positions, keys, self.map_hw(lat_size), heads, self.eng.text_len,
                           loss_scale=amp_loss_scale, **ekw)
        lo, hi = float(scale_range[0]), float(scale_range[1])

        def step_scale(index, n_timesteps):
            return latent_scale * (lo + (hi - lo) * index / max(n_timesteps - 1, 1)) ** 0.5 / amp_loss_scale
        gs = GuidanceState(en, amp_loss_scale, float("-inf"), 1, max_index_step, kind="boxdiff", step_scale=step_scale)
        gs.step_params = (float(amp_loss_scale), float(latent_scale), lo, hi)      # jobs batched together must agree on these
        return gs

    # ------------------------------------------------------------------------------------------
    MAX_STATES = 12

    STATE_MIN_STEPS = 50      # history capacity of a fresh state: any schedule up to the reference's default fits

    def _state(self, nb, C, L, T) -> _State:
        """Device state + captured graphs per (batch bucket, latent shape), with room for T steps: the step count is a
        CAPACITY of the history / coefficient tables, not part of the identity, so the graphs captured during a short
        run (bench.py's 2-step pre-build before the timed region) are the ones a 50-step run replays.  A longer
        schedule than the capacity re-creates the state (and its graphs).  Least recently used entries are dropped so
        a long run over many shapes keeps a bounded footprint."""
        key = (nb, C, L)
        st = self._states.pop(key, None)
        if st is None or st.ctab.shape[0] < T:
            st = _State(self.dev, nb, C, L, max(T, self.STATE_MIN_STEPS))
        self._states[key] = st                                  # (re)insert at the MRU end
        while len(self._states) > self.MAX_STATES:
            self._states.pop(next(iter(self._states)))
        return st

    def _runner(self, st: _State, name, fn):
        """fn enqueued eagerly or as a cached hipGraph."""
        if not self.use_graphs:
            return fn
        if name not in st.graphs:
            st.graphs[name] = HipGraph(fn)
        return st.graphs[name]

    def _guide_runners(self, st: _State, nb: int, L: int, fuser: bool, gkeys):
        """(plan, forward runner, backward+update runner) of the guidance pass for a batch of nb images:
        B=nb grad plan on the conditional text (second half of the text batch) and the zero-masked GLIGEN
        half (pipelines.py:381-384)."""
        eng = self.eng
        pg = eng.plan(nb, L, grad=True, fuser=fuser, stop_key=eng.last_key(gkeys), save_keys=gkeys,
                      text_batch_offset=nb, obj_batch_offset=0)

        def g_fwd():
            pg.forward(st.lat)

        def g_bwd():
            grad = pg.backward(self.grad_scale)
            ops.axpy(grad, st.lat, st.gtab, eng.dyn, 0, active=st.active)       # pipelines.py:62-69
        name = ("guide", fuser, tuple(gkeys))
        return pg, self._runner(st, name + ("fwd",), g_fwd), self._runner(st, name + ("bwd",), g_bwd)

    def _count(self, kind, fuser, nb):
        k = (kind, bool(fuser), int(nb))
        self.pass_counts[k] = self.pass_counts.get(k, 0) + 1

    def backward_guidance(self, states: List[Optional[GuidanceState]], energy: EnergyTables, plan_g, index: int,
                          st: _State, fwd_run, bwd_run, trace: Optional[list] = None, fuser: bool = False):
        """latent_backward_guidance (pipelines.py:16-82) for a batch of images on st.lat.  Each image keeps
        its own `while loss/scale > thr and it < max_iter` exit: an image that left the loop is masked out of
        the latent update (st.active) and its carried loss is not refreshed."""
        nb = len(states)
        energy.bind(plan_g.maps, plan_g.gmaps)
        it = 0
        while True:
            GATE.checkpoint()
            act = [gs is not None and index < gs.max_index_step and it < gs.iters_at(index)
                   and gs.loss / gs.loss_scale > gs.loss_threshold for gs in states]
            if not any(act):
                break
            st.active.copy_(torch.tensor([1.0 if a else 0.0 for a in act]))
            fwd_run()                                               # grad-plan forward from st.lat
            loss_dev = energy.run(self.eng.dyn, grad_scale=self.grad_scale)
            bwd_run()                                               # backward + masked latent update
            self._count("guide", fuser, nb)
            losses = loss_dev.tolist()                              # host sync, as pipelines.py:30
            for j, gs in enumerate(states):
                if act[j]:
                    gs.loss = losses[j]
                    gs.iterations += 1
                    gs.iterations_fuser_on += int(bool(fuser))
                    self.stats["guidance_iters"] += 1
            if trace is not None:
                trace.append(dict(index=index, it=it, loss=losses[0], losses=losses,
                                  grad=plan_g.g_latents.clone()))
            it += 1

    # ------------------------------------------------------------------------------------------
    def guidance_only(self, latents: torch.Tensor, cond_embeddings: torch.Tensor, num_inference_steps: int,
                      index: int, guidance: dict, *, gligen=None, fuser: bool = False,
                      trace: Optional[list] = None):
        """One latent_backward_guidance call (pipelines.py:16-82) at step `index` of a T-step schedule.
        Returns (latents, loss, state) like the reference."""
        eng, sch, dev = self.eng, self.scheduler, self.dev
        _, C, L, _ = latents.shape
        T = num_inference_steps
        st = self._state(1, C, L, T)
        sch.set_timesteps(T)
        st.gtab[:T].copy_(sch.guidance_step_table(dev))
        g = dict(guidance)
        gkeys = [tuple(k) for k in (g.get("guidance_attn_keys") or DEFAULT_GUIDANCE_ATTN_KEYS)]
        gs = g.pop("state", None) or self.make_guidance(L, g.pop("bboxes"), g.pop("object_positions"), **g)
        if gs is not None and gs.kind == "boxdiff":
            gkeys = [tuple(k) for k in (g.get("guidance_attn_keys") or BOXDIFF_GUIDANCE_ATTN_KEYS)]
            st.gtab[:T, 0].copy_(torch.tensor([gs.step_scale(i, T) for i in range(T)], dtype=F32))
        eng.prepare_timesteps([int(t) for t in sch.timesteps])
        cond = cond_embeddings.to(dev)
        eng.prepare_text(torch.cat([torch.zeros_like(cond), cond]))
        eng.set_step(index)
        pg, gf, gb = self._guide_runners(st, 1, L, fuser, gkeys)
        if gligen is not None:
            eng.prepare_gligen(boxes=gligen[0], positive_embeddings=gligen[1], masks=gligen[2])
        st.lat.copy_(latents.to(dev, F32))
        if gs is not None:
            self.backward_guidance([gs], gs.energy, pg, index, st, gf, gb, trace, fuser=fuser)
        return st.lat.clone(), (gs.loss if gs is not None else None), gs

    # ------------------------------------------------------------------------------------------
    def profile_passes(self, L: int, T: int, gligen: bool, main_batches=(1,), guide_batches=(1,), guidance_keys=None,
                       ratio_energy: bool = False):
        """Eager (non-graph) launch sequences of the plans the sampler replays, for per-kernel HIP-event
        timing by bench.py: [(kind, fuser, images, callable)].  Uses whatever run constants are loaded.
        ratio_energy: the layout-guidance baseline's energy (ratio branch, no reference maps) instead of LMD / LMD+'s."""
        eng = self.eng
        keys = [tuple(k) for k in (guidance_keys or DEFAULT_GUIDANCE_ATTN_KEYS)]
        plan_keys = sorted({OBJ_KEY_DEFAULT, *DEFAULT_GUIDANCE_ATTN_KEYS})
        out = []
        for f in ([True, False] if gligen else [False]):
            for nb in main_batches:
                plan = eng.plan(2 * nb, L, fuser=f, save_keys=plan_keys)
                st_m = self._state(nb, eng.cfg.in_channels, L, T)

                def main(plan=plan, st=st_m, nb=nb):                       # as _denoise_chunk's main_fn
                    ops.copy_(plan.latents_in[:nb], st.lat)
                    ops.copy_(plan.latents_in[nb:], st.lat)
                    plan.forward()
                    ops.cfg_ddim_step(plan.eps_out, st.lat, st.lat, st.ctab, eng.dyn, frozen_ref=st.frozen_ref,
                                      mask=st.mask, hist=st.hist)
                out.append(("main", f, nb, main))
            for nb in guide_batches:
                st = self._state(nb, eng.cfg.in_channels, L, T)
                pg = eng.plan(nb, L, grad=True, fuser=f, stop_key=eng.last_key(keys), save_keys=keys,
                              text_batch_offset=nb)

                # the energy launch between the two halves: a canonical two-box layout per image (its cost depends on
                # the item count only, 2 boxes x 3 tokens x 4 keys + reference terms, not on the boxes)
                boxes = [[0.15, 0.35, 0.5, 0.8], [0.6, 0.38, 0.98, 0.8]]
                refs = None if ratio_energy else torch.full((T, 2, len(keys), self.heads_of(keys[0]),
                                                             max(self.map_hw(L)[k] for k in keys)), 1e-3, device=self.dev)
                en = [self.make_guidance(L, [[b] for b in boxes], [[1, 2, 3], [5, 6, 7]], guidance_attn_keys=keys,
                                         use_ratio_based_loss=ratio_energy, word_token_indices=[3, 7],
                                         ref_ca_word_token_only=True, ref_maps=refs, max_index_step=1).energy
                      for _ in range(nb)]
                energy = en[0] if nb == 1 else EnergyTables.merged(en)
                energy.bind(pg.maps, pg.gmaps)

                def guide(pg=pg, st=st, energy=energy):
                    pg.forward(st.lat)
                    energy.run(eng.dyn, grad_scale=self.grad_scale)
                    pg.backward(self.grad_scale)
                    ops.axpy(pg.g_latents, st.lat, st.gtab, eng.dyn, 0, active=st.active)
                out.append(("guide", f, nb, guide))
        return out

    # ------------------------------------------------------------------------------------------
    def denoise(self, latents, text_embeddings, num_inference_steps, *, gligen=None, guidance=None,
                frozen_mask=None, return_token_ca_only=None, **shared):
        """Single-image form of denoise_batch (same arguments as before)."""
        job = Job(latents, text_embeddings, gligen, guidance, frozen_mask, return_token_ca_only)
        return self.denoise_batch([job], num_inference_steps, use_gligen=gligen is not None, **shared)[0]

    def denoise_batch(self, jobs: List[Job], num_inference_steps: int, **kw):
        """Any number of independent images: guided and unguided jobs are chunked separately (at most
        max_batch_guided / max_batch images per UNet call — the reference runs them one at a time, so any count
        works there too), every chunk is padded to a bucket size with inert copies of its last job, and the
        results come back in job order.  Arguments and per-job result as `_denoise_chunk`."""
        out: List[Optional[dict]] = [None] * len(jobs)
        for guided, cap in ((True, self.max_batch_guided), (False, self.max_batch)):
            idx = [i for i, j in enumerate(jobs) if (j.guidance is not None) == guided]
            c0 = 0
            for count, nb in plan_chunks(len(idx), cap, self.BUCKETS, self.max_pad):
                part = idx[c0:c0 + count]
                c0 += count
                chunk = [jobs[i] for i in part]
                last = chunk[-1]
                pad = [Job(last.latents, last.text, last.gligen, None, last.frozen_mask, last.token)
                       for _ in range(nb - len(chunk))]
                self.stats["images"] += len(chunk)
                self.stats["padded_images"] += len(pad)
                res = self._denoise_chunk(chunk + pad, num_inference_steps, **kw)
                for i, r in zip(part, res):
                    out[i] = r
        return out

    def _denoise_chunk(self, jobs: List[Job], num_inference_steps: int, *, guidance_scale: float = 7.5,
                       use_gligen: bool = False, gligen_scheduled_sampling_beta: float = 0.3,
                       frozen_steps: int = 0, saved_cross_attn_keys: Sequence[Tuple] = (),
                       return_cond_ca_only: bool = False, save_all_latents: bool = True,
                       trace: Optional[list] = None, fast_after_steps: Optional[int] = None, fast_rate: int = 2,
                       first_step: int = 0, n_steps: Optional[int] = None):
        """Generic 50-step loop over a batch of independent images (one UNet call serves all of them:
        B = 2*len(jobs) for the CFG pass, len(jobs) for the guidance pass).

        job.latents: (1,C,L,L) start latents or (T+1,1,C,L,L) history whose [0] is the start and whose
          [i+1] feeds the frozen-mask blend (pipelines.py:340-345, 445-446, 585-586).
        job.text: (2,77,Cx) = [uncond; cond];  job.gligen: (boxes (2,30,4), embeddings (2,30,768), masks (2,30));
        job.guidance: dict(bboxes, object_positions, **semantic_guidance_kwargs, [ref_maps]) or None;
        job.token: return_token_ca_only.
        fast_after_steps / fast_rate: the optional fast tail (pipelines.py:151-152,358-359 + schedule.py):
          after that many steps only every fast_rate-th timestep is run, with the DDIM step size re-derived
          per step (dynamic_num_inference_steps, lmd_plus.py:109); T_run < T steps are executed.
        first_step / n_steps: run only steps first_step .. first_step + n_steps - 1 of the schedule, starting from
          the given latents (which then stand for the state BEFORE step first_step).  Used by the teacher-forced
          parity tests (one guided step from the reference's own latents of that step) and by partial schedules.
        Returns per job dict(latents (1,C,L,L), latents_all (T_run+1,1,C,L,L), saved {key: [T_run,Bp,H,HW,Tp]},
        guidance_iters).
        """
        eng, sch, dev = self.eng, self.scheduler, self.dev
        nb = len(jobs)
        starts = [j.latents[0] if j.latents.dim() == 5 else j.latents for j in jobs]
        _, C, L, _ = starts[0].shape
        T = num_inference_steps
        st = self._state(nb, C, L, T)
        sch.set_timesteps(T)
        ts = sch.timesteps
        if fast_after_steps is not None:
            ts = sch.fast_schedule(ts, int(fast_after_steps), int(fast_rate))
        Tr = len(ts)                                                          # steps actually run
        multistep = bool(getattr(sch, "multistep", False))
        if multistep:
            if fast_after_steps is not None:
                raise RuntimeError("the fast schedule (utils/schedule.py) re-derives DDIM step sizes; not defined for the "
                                   "multistep scheduler")
            st.mtab[:Tr].copy_(sch.multistep_table(guidance_scale, dev, timesteps=ts))
        else:
            st.ctab[:Tr].copy_(sch.coef_table(guidance_scale, dev, timesteps=ts, step_ratios=sch.dynamic_step_sizes(ts)))
        st.gtab[:Tr].copy_(sch.guidance_step_table(dev, timesteps=ts))
        n_ground = int(gligen_scheduled_sampling_beta * Tr) if use_gligen else 0   # pipelines.py:405
        save_keys = [tuple(k) for k in saved_cross_attn_keys]
        # a superset of keys is always captured by the main plans so that one graph serves every caller
        plan_keys = sorted(set(save_keys) | {OBJ_KEY_DEFAULT, *DEFAULT_GUIDANCE_ATTN_KEYS}) if save_keys else []

        def fuser_at(index):
            return bool(use_gligen and index < n_ground)                     # pipelines.py:408-414
        gstates: List[Optional[GuidanceState]] = []
        gkeys = None
        for j in jobs:
            if j.guidance is None:
                gstates.append(None)
                continue
            g = dict(j.guidance)
            jk = [tuple(k) for k in (g.get("guidance_attn_keys") or
                                     (BOXDIFF_GUIDANCE_ATTN_KEYS if g.get("use_boxdiff") else DEFAULT_GUIDANCE_ATTN_KEYS))]
            if gkeys is not None and jk != gkeys:
                raise RuntimeError("jobs guided in one batch must share guidance_attn_keys")
            gkeys = jk
            gstates.append(self.make_guidance(L, g.pop("bboxes"), g.pop("object_positions"), **g))
        guided = any(gs is not None for gs in gstates)
        energy = None
        if guided:
            kinds = {gs.kind for gs in gstates if gs is not None}
            if len(kinds) != 1:
                raise RuntimeError("jobs guided in one batch must share the energy (LMD guidance or BoxDiff)")
            tables = BoxDiffTables if kinds == {"boxdiff"} else EnergyTables
            if kinds == {"boxdiff"}:             # utils/boxdiff.py:236-238: the update's own step table (all jobs alike)
                gs0 = next(gs for gs in gstates if gs is not None)
                if any(gs is not None and getattr(gs, "step_params", None) != getattr(gs0, "step_params", None) for gs in gstates):
                    raise RuntimeError("BoxDiff jobs guided in one batch must share amp_loss_scale / latent_scale / scale_range")
                st.gtab[:Tr, 0].copy_(torch.tensor([gs0.step_scale(i, Tr) for i in range(Tr)], dtype=F32))
            energy = gstates[[gs is not None for gs in gstates].index(True)].energy if nb == 1 else \
                tables.merged([gs.energy if gs is not None else None for gs in gstates])
        max_guided = max([gs.max_index_step for gs in gstates if gs is not None] + [0])

        # ---- per-run constants (before graph capture so that warm-up launches see valid inputs)
        eng.prepare_timesteps([int(t) for t in ts])
        eng.prepare_text(torch.cat([j.text[0:1] for j in jobs] + [j.text[1:2] for j in jobs]))
        eng.set_step(0)
        eng.dyn[1:2].fill_(0)

        # ---- plans + launch sequences (built/captured once per shape, cached)
        runners_main, plans_main, runners_guide = {}, {}, {}
        for f in {fuser_at(i) for i in range(Tr)}:
            plan = plans_main[f] = eng.plan(2 * nb, L, fuser=f, save_keys=plan_keys)

            def main_fn(plan=plan):
                ops.copy_(plan.latents_in[:nb], st.lat)                      # torch.cat([latents]*2)
                ops.copy_(plan.latents_in[nb:], st.lat)
                plan.forward()
                if multistep:
                    ops.cfg_multistep_step(plan.eps_out, st.lat, st.lat, st.x0_prev, st.mtab, eng.dyn,
                                           frozen_ref=st.frozen_ref, mask=st.mask, hist=st.hist)
                else:
                    ops.cfg_ddim_step(plan.eps_out, st.lat, st.lat, st.ctab, eng.dyn, frozen_ref=st.frozen_ref,
                                      mask=st.mask, hist=st.hist)
            runners_main[f] = (main_fn, ("main", f, tuple(plan_keys), multistep))
        if guided:
            for f in {fuser_at(i) for i in range(min(max_guided, Tr))}:
                runners_guide[f] = self._guide_runners(st, nb, L, f, gkeys)
        if use_gligen:                                                        # after the plans exist
            eng.prepare_gligen(boxes=torch.cat([j.gligen[0][0:1] for j in jobs] + [j.gligen[0][1:2] for j in jobs]),
                               positive_embeddings=torch.cat([j.gligen[1][0:1] for j in jobs] +
                                                             [j.gligen[1][1:2] for j in jobs]),
                               masks=torch.cat([j.gligen[2][0:1] for j in jobs] + [j.gligen[2][1:2] for j in jobs]))
        runners_main = {f: self._runner(st, name, fn) for f, (fn, name) in runners_main.items()}

        # ---- state of this call
        st.lat.copy_(torch.cat([s.to(dev, F32) for s in starts]))
        first_step = max(0, min(int(first_step), Tr))
        last_step = Tr if n_steps is None else min(Tr, first_step + int(n_steps))
        if multistep and first_step > 0:
            # row `first_step` of a second-order schedule mixes in the data prediction of step first_step - 1, which a
            # run that starts here does not have (st.x0_prev holds whatever run used this state last)
            raise RuntimeError("partial schedules (first_step > 0) are not defined for the multistep scheduler")
        st.hist[first_step].copy_(st.lat)
        st.mask.zero_()
        if frozen_steps > 0:
            for b, j in enumerate(jobs):
                if j.frozen_mask is not None and j.latents.dim() == 5:
                    rows = min(j.latents.shape[0], st.frozen_ref.shape[0])   # a fast-schedule history is shorter
                    st.frozen_ref[:rows, b].copy_(j.latents[:rows, 0].to(dev, F32))
                    st.mask[b].copy_(j.frozen_mask.to(dev, F32).clamp(0., 1.).reshape(L * L))
            eng.dyn[1:2].fill_(int(frozen_steps))
        hw = self.map_hw(L)
        Bp = 1 if return_cond_ca_only else 2
        saved = [{k: torch.zeros((Tr, Bp, self.heads_of(k), hw[k], 1 if j.token is not None else eng.text_len),
                                 device=dev, dtype=F32) for k in save_keys} for j in jobs]

        for index in range(first_step, last_step):
            GATE.checkpoint()                                            # lanes.py: another lane may be waiting to capture
            eng.set_step(index)
            fuser_on = fuser_at(index)
            if guided and index < max_guided:
                pg, gf, gb = runners_guide[fuser_on]
                self.backward_guidance(gstates, energy, pg, index, st, gf, gb, trace, fuser=fuser_on)
            runners_main[fuser_on]()
            self.stats["unet_main"] += 1
            self._count("main", fuser_on, nb)
            if save_keys:
                maps = plans_main[fuser_on].maps
                for b, j in enumerate(jobs):                              # attention_processor.py:466-476
                    for k in save_keys:
                        m = maps[k][nb + b:nb + b + 1] if return_cond_ca_only else maps[k][[b, nb + b]]
                        saved[b][k][index].copy_(m[..., int(j.token):int(j.token) + 1] if j.token is not None else m)
        hist = st.hist[:Tr + 1].clone() if save_all_latents else None
        return [dict(latents=st.lat[b:b + 1].clone(), latents_all=hist[:, b:b + 1] if hist is not None else None,
                     saved=saved[b], guidance_iters=gstates[b].iterations if gstates[b] is not None else 0,
                     guidance_iters_fuser_on=gstates[b].iterations_fuser_on if gstates[b] is not None else 0)
                for b in range(nb)]

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, latents: torch.Tensor):
        """pipelines.py:117-127: VAE decode -> uint8 HWC (the VAE is [ext] and stays PyTorch/MIOpen)."""
        if self.vae is None:
            raise RuntimeError("no VAE attached to the sampler")
        outs = []
        for c0 in range(0, latents.shape[0], 8):                  # bounded activation footprint
            image = self.vae.decode(latents[c0:c0 + 8] / 0.18215)
            image = (image / 2 + 0.5).clamp(0, 1)
            outs.append((image.detach().float().permute(0, 2, 3, 1) * 255).round().to(torch.uint8))
        return torch.cat(outs).cpu().numpy()
