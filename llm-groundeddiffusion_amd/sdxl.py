"""SDXL-refiner post-pass on the HIP engine (BASELINE config 5).

The reference's whole implementation is one third-party call (generation/sdxl_refinement.py:13-15,24-29):

    pipe = StableDiffusionXLImg2ImgPipeline.from_pretrained("stabilityai/stable-diffusion-xl-refiner-1.0", fp16)
    image = Image.fromarray(image).resize((1024, 1024), Image.LANCZOS)
    output = pipe(overall_prompt, image=image, negative_prompt=extra_neg + ", " + sdxl_negative_prompt,
                  strength=refinement_step_ratio, generator=torch.manual_seed(refine_seed)).images[0]

`SDXLRefiner.refine()` is that `pipe(...)` call ([ext] diffusers >= 0.19 StableDiffusionXLImg2ImgPipeline.__call__ with
its defaults: 50 scheduler steps of which the last int(50 * strength) run, guidance_scale 5.0, aesthetic scores 6.0 / 2.5,
original size = target size = the image, crop (0, 0)), on this package's kernels:

    image -> HipVAEEncoder (posterior mean / logvar) -> sample with the caller's generator -> x 0.13025 -> add_noise
    -> [ scale_model_input -> UNet (CFG pair: negative, positive; text_time conditioning) -> CFG + Euler step ] x N
    -> / 0.13025 -> HipVAEDecoder -> uint8

The UNet is the same `UNetEngine` as the SD path with the refiner's configuration (`weights.CONFIGS["sdxl_refiner"]`:
attention only at the two middle resolutions, four transformer layers per block, 64-wide heads, pooled-text + size / score
conditioning added to the time embedding per image); one captured hipGraph = scale + UNet + fused CFG / Euler update
replays for every step.  diffusers is absent from the sandbox — parity unpinned at that boundary (oracle/restate_sdxl.py
says what IS pinned).
"""
from typing import Optional

import numpy as np
import torch

from . import ops
from .scheduler import EulerDiscreteScheduler
from .unet import UNetEngine

F16, F32 = torch.float16, torch.float32
SDXL_VAE_SCALING = 0.13025            # [ext] sdxl-vae config.json scaling_factor


def add_time_ids(height, width, aesthetic_score=6.0, negative_aesthetic_score=2.5):
    """[ext] StableDiffusionXLImg2ImgPipeline._get_add_time_ids with requires_aesthetics_score (the refiner):
    (original h, w, crop top, left, score); rows ordered like the CFG batch: negative, positive."""
    return torch.tensor([[float(height), float(width), 0.0, 0.0, float(negative_aesthetic_score)],
                         [float(height), float(width), 0.0, 0.0, float(aesthetic_score)]], dtype=F32)


class SDXLRefiner:
    def __init__(self, engine: UNetEngine, vae_encoder, vae_decoder, text_encoder=None, tokenizer=None,
                 scheduler: Optional[EulerDiscreteScheduler] = None, scaling_factor: float = SDXL_VAE_SCALING,
                 use_graphs: bool = True):
        if engine.cfg.addition_embed_type != "text_time":
            raise ValueError(f"{engine.cfg.name}: the refiner needs a text_time UNet configuration")
        self.eng = engine
        self.dev = engine.device
        self.enc, self.dec = vae_encoder, vae_decoder
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.scheduler = scheduler or EulerDiscreteScheduler()
        self.scaling_factor = float(scaling_factor)
        self.use_graphs = use_graphs
        self._state = {}

    # ---- text ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_prompt(self, prompt: str, negative_prompt: str):
        """[ext] encode_prompt of the refiner (one tokenizer / text tower: OpenCLIP ViT-bigG/14): conditioning =
        hidden_states[-2], pooled = text_embeds; returns ([2, 77, Cx], [2, P]) ordered (negative, positive)."""
        if self.text_encoder is None or self.tokenizer is None:
            raise RuntimeError("no tokenizer / text encoder attached: pass prompt_embeds and pooled embeddings")
        ids = self.tokenizer([negative_prompt, prompt], padding="max_length", max_length=self.tokenizer.model_max_length,
                             truncation=True, return_tensors="pt").input_ids
        out = self.text_encoder(ids.to(self.dev), output_hidden_states=True)
        return out.hidden_states[-2], out[0]

    # ---- device state of one (latent size, step count) ------------------------------------------------------------
    def _get_state(self, L, n_steps):
        key = (L, n_steps)
        if key not in self._state:
            C = self.eng.cfg.in_channels
            st = dict(lat=torch.zeros((1, C, L, L), device=self.dev, dtype=F32),
                      x0_prev=torch.zeros((1, C, L, L), device=self.dev, dtype=F32),
                      tab=torch.zeros((n_steps, 8), device=self.dev, dtype=F32), graph=None)
            self._state = {key: st}              # one resident size at a time
        return self._state[key]

    @torch.no_grad()
    def refine_latents(self, latents, prompt_embeds, pooled, *, first_index: int, num_inference_steps: int = 50,
                       guidance_scale: float = 5.0, height: int = 1024, width: int = 1024, aesthetic_score: float = 6.0,
                       negative_aesthetic_score: float = 2.5, trace: Optional[list] = None):
        """The denoising loop of the pipeline from scheduler step `first_index` on.  latents [1, 4, L, L] already
        noised to sigma[first_index]; prompt_embeds [2, 77, Cx], pooled [2, P] ordered (negative, positive)."""
        eng, sch = self.eng, self.scheduler
        sch.set_timesteps(num_inference_steps)
        ts = [float(t) for t in sch.timesteps[first_index:]]
        n = len(ts)
        L = latents.shape[-1]
        st = self._get_state(L, n)
        st["lat"].copy_(latents.to(self.dev, F32))
        st["tab"].copy_(sch.multistep_table(guidance_scale, self.dev, first=first_index))
        eng.prepare_text(prompt_embeds)
        eng.prepare_timesteps(ts, dict(text_embeds=pooled, time_ids=add_time_ids(height, width, aesthetic_score,
                                                                                   negative_aesthetic_score)))
        plan = eng.plan(2, L)
        lat, tab, x0p = st["lat"], st["tab"], st["x0_prev"]

        def one_step():
            # scale_model_input for the (negative, positive) pair, UNet, classifier-free guidance + Euler update
            ops.scale_rows(lat, plan.latents_in, tab, eng.dyn, EulerDiscreteScheduler.C_IN_COL, reps=2)
            plan.forward()
            ops.cfg_multistep_step(plan.eps_out, lat, lat, x0p, tab, eng.dyn)
        if self.use_graphs and st["graph"] is None:
            from .sampler import HipGraph
            eng.set_step(0)
            keep = lat.clone()
            st["graph"] = HipGraph(one_step)     # the capture's warm-up run advanced the latents: restore them
            lat.copy_(keep)
        run = st["graph"] if self.use_graphs else one_step
        from .lanes import GATE
        for i in range(n):
            GATE.checkpoint()                    # a safe point per step: another lane's graph capture may proceed
            eng.set_step(i)
            run()
            if trace is not None:
                trace.append(lat.clone())
        return lat.clone()

    @torch.no_grad()
    def prepare_latents(self, image, seed: int, first_timestep: float):
        """[ext] prepare_latents: posterior sample of the VAE encoder with the caller's CPU generator
        (sdxl_refinement.py:25: `g = torch.manual_seed(refine_seed)`), scaled, then add_noise at the first timestep.
        The pipeline draws the posterior noise in the VAE's dtype (fp32: config.force_upcast) and the diffusion noise in
        the prompt embeddings' dtype (fp16) — a CPU fp16 draw consumes the generator differently from an fp32 one."""
        from .hostprep import RNG_LOCK
        mean, logvar = self.enc.encode_moments(image)
        with RNG_LOCK:
            g = torch.manual_seed(int(seed))
            e1 = torch.randn(mean.shape, generator=g, dtype=F32)
            e2 = torch.randn(mean.shape, generator=g, dtype=F16)
        lat = (mean + torch.exp(0.5 * logvar) * e1.to(self.dev)) * self.scaling_factor
        return self.scheduler.add_noise(lat, e2.to(self.dev, F32), first_timestep)

    @torch.no_grad()
    def refine(self, image, prompt_embeds=None, pooled=None, *, prompt: Optional[str] = None,
               negative_prompt: Optional[str] = None, seed: int = 0, strength: float = 0.3,
               num_inference_steps: int = 50, guidance_scale: float = 5.0, output: str = "uint8"):
        """image: uint8 [H, W, 3] (numpy) or float [1, 3, H, W] in [-1, 1]; H = W, a multiple of 8 (the reference
        resizes to 1024 x 1024 first).  Returns uint8 [H, W, 3] (output="uint8"), the float image [1, 3, H, W] in
        [-1, 1] before clamping ("float") or the final latents ("latent")."""
        if prompt_embeds is None:
            prompt_embeds, pooled = self.encode_prompt(prompt, negative_prompt or "")
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1)[None].float() / 255.0 * 2.0 - 1.0
        H, W = image.shape[-2:]
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        first = sch.img2img_start(num_inference_steps, strength)
        if first >= num_inference_steps:
            raise ValueError(f"strength {strength} leaves no denoising step")
        lat = self.prepare_latents(image, seed, float(sch.timesteps[first]))
        if not bool(torch.isfinite(lat).all()):
            # fp16 activations of the HIP VAE encoder overflowed (the real SDXL VAE needs fp32: config.force_upcast)
            raise RuntimeError("non-finite latents from the VAE encoder (fp16 overflow; this VAE needs an fp32 / fp16-safe variant)")
        lat = self.refine_latents(lat, prompt_embeds, pooled, first_index=first, num_inference_steps=num_inference_steps,
                                  guidance_scale=guidance_scale, height=H, width=W)
        if output == "latent":
            return lat
        img = self.dec.decode(lat / self.scaling_factor)
        if not bool(torch.isfinite(img).all()):
            raise RuntimeError("non-finite image from the VAE decoder (fp16 overflow; this VAE needs an fp32 / fp16-safe variant)")
        if output == "float":
            return img
        return ((img[0] / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy()


def build_synthetic(config="sdxl_refiner", device="cuda", seed=0, vae_ch=(128, 256, 512, 512), vae_layers=2):
    """The refiner with seeded random parameters of the real architectures (no checkpoints in the sandbox):
    UNet `config`, SD / SDXL VAE encoder + decoder.  No text tower (callers pass embeddings)."""
    from . import vae, weights
    cfg = weights.CONFIGS[config]
    eng = UNetEngine(cfg, device, weights.synth_state_dict(cfg, seed), max_text_batch=2)
    sd = vae.synth_aekl_state_dict(vae_ch, vae_layers, seed=seed)
    return SDXLRefiner(eng, vae.HipVAEEncoder(sd, device), vae.HipVAEDecoder(sd, device)), sd
