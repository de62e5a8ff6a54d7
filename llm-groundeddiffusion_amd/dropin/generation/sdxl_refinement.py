"""Drop-in for generation/sdxl_refinement.py (the `--sdxl` post-pass of generate.py:222-224,383-384): same module
surface — `sdxl_negative_prompt`, `pipe`, `init(offload_model)`, `refine(image, spec, refine_seed,
refinement_step_ratio)` — with the img2img pass running on the HIP engine (`lgd_amd.sdxl.SDXLRefiner`).

`init()` needs the refiner checkpoint, which the reference takes from the Hugging Face hub through diffusers
(sdxl_refinement.py:13-15); where diffusers and the checkpoint are available it is loaded once on the host and every
network is re-hosted on the HIP kernels (UNet -> UNetEngine, VAE -> HipVAEEncoder / HipVAEDecoder, text_encoder_2 ->
HipCLIPTextEncoder); `offload_model` has nothing to offload (288 GB of HBM) and is accepted for call compatibility.
Offline (this sandbox, the benchmark) `init_synthetic()` builds the same networks with seeded random parameters and
`refine()` then takes the prompt embeddings from `spec["sdxl_prompt_embeds"]` / `spec["sdxl_pooled"]`."""
import os as _os

import numpy as np
import torch
from PIL import Image

from lgd_amd import sdxl as _sdxl

# This is adapted to SDXL since it often generates styles that we don't want (sdxl_refinement.py:5-6).
sdxl_negative_prompt = "drawing, painting, crayon, sketch, graphite, impressionist, noisy, blurry, soft, deformed, ugly"

pipe = None
REFINE_SIZE = 1024                     # sdxl_refinement.py:26


def init(offload_model=True, device="cuda"):
    """sdxl_refinement.py:10-22."""
    global pipe
    try:
        from diffusers import StableDiffusionXLImg2ImgPipeline
    except ImportError as e:
        raise RuntimeError("generation.sdxl_refinement.init() loads stabilityai/stable-diffusion-xl-refiner-1.0 through "
                           "diffusers, which is not installed here; use init_synthetic() for seeded random weights") from e
    hf = StableDiffusionXLImg2ImgPipeline.from_pretrained("stabilityai/stable-diffusion-xl-refiner-1.0",
                                                          torch_dtype=torch.float16, variant="fp16", use_safetensors=True)
    pipe = from_diffusers(hf, device)
    return pipe


def from_diffusers(hf_pipe, device="cuda"):
    """A loaded StableDiffusionXLImg2ImgPipeline -> SDXLRefiner: weights are read from the modules' state dicts, the
    UNet configuration from `unet.config` (must be the refiner's topology family: text_time, linear projections)."""
    from lgd_amd import clip, vae, weights
    from lgd_amd.scheduler import EulerDiscreteScheduler
    from lgd_amd.unet import UNetEngine
    c = hf_pipe.unet.config
    depth = c.transformer_layers_per_block if isinstance(c.transformer_layers_per_block, int) else c.transformer_layers_per_block[0]
    cfg = weights.UNetConfig(name="sdxl_refiner_ckpt", in_channels=c.in_channels, out_channels=c.out_channels,
                             block_out_channels=tuple(c.block_out_channels), layers_per_block=c.layers_per_block,
                             cross_attention_dim=c.cross_attention_dim, attention_head_dim=tuple(c.attention_head_dim),
                             norm_num_groups=c.norm_num_groups, norm_eps=c.norm_eps,
                             use_linear_projection=c.use_linear_projection, sample_size=c.sample_size,
                             down_attn=tuple(t.startswith("CrossAttn") for t in c.down_block_types),
                             up_attn=tuple(t.startswith("CrossAttn") for t in c.up_block_types),
                             transformer_depth=depth, addition_embed_type=c.addition_embed_type,
                             addition_time_embed_dim=c.addition_time_embed_dim,
                             projection_class_embeddings_input_dim=c.projection_class_embeddings_input_dim)
    eng = UNetEngine(cfg, device, {k: v.float().cpu() for k, v in hf_pipe.unet.state_dict().items()}, max_text_batch=2)
    vsd = hf_pipe.vae.state_dict()
    if getattr(hf_pipe.vae.config, "force_upcast", False) and not _os.environ.get("LGD_SDXL_VAE_FP16"):
        # The SDXL VAE is exported with force_upcast=true: its activations overflow fp16, so the reference pipeline runs
        # encode and decode in fp32.  HipVAEEncoder / HipVAEDecoder keep every activation in fp16 -> inf / NaN latents
        # or black images with the real checkpoint.  Refuse instead of producing them silently; a checkpoint with the
        # fp16-safe VAE weights (config.force_upcast = false) loads, and LGD_SDXL_VAE_FP16=1 overrides (refine() then
        # still checks that the moments and the decoded image are finite).
        raise RuntimeError("this VAE sets config.force_upcast (fp16 overflow): the fp16 HIP VAE would produce non-finite "
                           "latents; load an fp16-safe VAE (force_upcast=false) or set LGD_SDXL_VAE_FP16=1 to try anyway")
    sch = hf_pipe.scheduler.config
    return _sdxl.SDXLRefiner(eng, vae.HipVAEEncoder(vsd, device), vae.HipVAEDecoder(vsd, device),
                             text_encoder=clip.from_hf(hf_pipe.text_encoder_2, device), tokenizer=hf_pipe.tokenizer_2,
                             scheduler=EulerDiscreteScheduler(sch.num_train_timesteps, sch.beta_start, sch.beta_end,
                                                              sch.steps_offset, sch.prediction_type),
                             scaling_factor=hf_pipe.vae.config.scaling_factor)


def init_synthetic(config="sdxl_refiner", device="cuda", seed=0):
    """Offline stand-in for init(): the refiner's networks with seeded random parameters."""
    global pipe
    pipe, _ = _sdxl.build_synthetic(config, device, seed)
    return pipe


def refine(image, spec, refine_seed, refinement_step_ratio=0.5):
    """sdxl_refinement.py:24-30.  image: uint8 [H, W, 3]; returns a PIL image like the reference."""
    if pipe is None:
        raise RuntimeError("call init() (or init_synthetic()) first")
    # sdxl_refinement.py:24-29: LANCZOS resize to 1024 x 1024, the layout's own negative prompt in front of the style list
    resized = np.asarray(Image.fromarray(image).resize((REFINE_SIZE,) * 2, Image.LANCZOS))
    text = dict(prompt=spec["prompt"], negative_prompt=", ".join([spec["extra_neg_prompt"], sdxl_negative_prompt]))
    if "sdxl_prompt_embeds" in spec:          # cached text side (no tokenizer / text tower offline)
        text.update(prompt_embeds=spec["sdxl_prompt_embeds"], pooled=spec["sdxl_pooled"])
    out = pipe.refine(resized, seed=refine_seed, strength=refinement_step_ratio, **text)
    return Image.fromarray(out)
