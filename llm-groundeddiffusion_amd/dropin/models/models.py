"""models/models.py of the reference: process-global model handles (`sd_key`, `sd_version`,
`model_dict`), prompt encoding and the loader.  `encode_prompts` / `process_input_embeddings` are boundary glue
whose behaviour the callers prescribe (tokenise, pad to 77, encode uncond + cond, concatenate): they follow
models/models.py:63-109 closely by necessity.  `model_dict.unet` is the HIP engine wrapped in the
reference's UNet2DConditionModel call surface."""
import torch

from lgd_amd import weights as _weights
from lgd_amd.sampler import LMDSampler
from lgd_amd.scheduler import DDIMScheduler, DPMSolverMultistepScheduler
from lgd_amd.unet import UNetEngine
from utils import torch_device  # noqa: F401

from .unet_2d_condition import UNet2DConditionModel

# set by generate.py (generate.py:104-123)
sd_key = ""
sd_version = ""
model_dict = None


class _EasyDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def build_model_dict(cfg, state_dict, vae=None, tokenizer=None, text_encoder=None, device="cuda",
                     dtype=torch.float16, scheduler_config=None, use_dpm_multistep_scheduler=False):
    """model_dict contract of models/models.py:55: vae, tokenizer, text_encoder, unet, scheduler, dtype.
    scheduler_config: the checkpoint's own scheduler_config.json fields (models/models.py:49 reads them through
    DDIMScheduler.from_pretrained); only what the DDIM eta=0 path uses is honoured, anything else must match."""
    eng = UNetEngine(cfg, device, state_dict)
    unet = UNet2DConditionModel(eng)
    sc = dict(scheduler_config or {})
    for k, want in (("beta_schedule", "scaled_linear"), ("clip_sample", False), ("set_alpha_to_one", False)):
        if k in sc and sc[k] != want:
            raise RuntimeError(f"scheduler_config.{k}={sc[k]!r} is not supported on the HIP path (expects {want!r})")
    if "prediction_type" in sc and sc["prediction_type"] != cfg.prediction_type:
        raise RuntimeError("scheduler prediction_type disagrees with the UNet config")
    skw = dict(num_train_timesteps=sc.get("num_train_timesteps", 1000), beta_start=sc.get("beta_start", 0.00085),
               beta_end=sc.get("beta_end", 0.012), prediction_type=cfg.prediction_type)
    if use_dpm_multistep_scheduler:                       # models/models.py:46-47
        sched = DPMSolverMultistepScheduler(**skw)
    else:
        sched = DDIMScheduler(steps_offset=sc.get("steps_offset", 1), **skw)
    md = _EasyDict(vae=vae, tokenizer=tokenizer, text_encoder=text_encoder, unet=unet, scheduler=sched, dtype=dtype)
    md["sampler"] = LMDSampler(eng, sched, vae=vae)
    return md


def load_synthetic(name="sd14_gligen", seed=0, device="cuda", with_vae=True):
    """Seeded random weights of the exact architecture (no checkpoints in the sandbox)."""
    from lgd_amd.vae import make_hip_vae
    cfg = _weights.CONFIGS[name]
    return build_model_dict(cfg, _weights.synth_state_dict(cfg, seed), vae=make_hip_vae(device) if with_vae else None,
                            device=device)


def load_sd(key="runwayml/stable-diffusion-v1-5", use_fp16=False, load_inverse_scheduler=False,
            use_dpm_multistep_scheduler=False, scheduler_cls=None):
    """models/models.py:16-62.  Needs the Hugging Face checkpoint (diffusers + network/cache); the UNet
    state dict is repacked into the HIP engine's arenas, the CLIP text tower runs on the HIP kernels
    (lgd_amd.clip), the VAE decoder's state dict is repacked for the HIP kernels (lgd_amd.vae.HipVAEDecoder;
    LGD_HF_VAE=1 in the environment keeps the Hugging Face module, for A/B checks against it)."""
    if not use_fp16:      # models/models.py:33-38: the reference then loads fp32 weights ("run final results in fp32")
        from generation._common import note_precision
        note_precision("models.load_sd", "use_fp16=False")
    try:
        from diffusers import AutoencoderKL, DDIMScheduler as HFDDIM, UNet2DConditionModel as HFUNet
        from transformers import CLIPTextModel, CLIPTokenizer
    except ImportError as e:
        raise RuntimeError("load_sd needs `diffusers` and HF checkpoints; use load_synthetic() offline") from e
    if scheduler_cls is not None:
        name = getattr(scheduler_cls, "__name__", str(scheduler_cls))
        if use_dpm_multistep_scheduler:
            raise AssertionError("`use_dpm_multistep_scheduler` cannot be used with `scheduler_cls`")     # models.py:51
        if name == "DPMSolverMultistepScheduler":
            use_dpm_multistep_scheduler = True
        elif name != "DDIMScheduler":
            raise RuntimeError(f"scheduler {name}: the HIP path implements DDIMScheduler and DPMSolverMultistepScheduler")
    hf = HFUNet.from_pretrained(key, subfolder="unet")
    c = hf.config
    heads = c.attention_head_dim if isinstance(c.attention_head_dim, (list, tuple)) else (c.attention_head_dim,) * 4
    sched_cfg = dict(HFDDIM.from_pretrained(key, subfolder="scheduler").config)      # models/models.py:49
    cfg = _weights.UNetConfig(name=key, block_out_channels=tuple(c.block_out_channels), cross_attention_dim=c.cross_attention_dim,
                              attention_head_dim=tuple(heads), use_linear_projection=getattr(c, "use_linear_projection", False),
                              use_gated_attention="gligen" in key, sample_size=c.sample_size,
                              prediction_type=sched_cfg.get("prediction_type", "epsilon"))
    vae = AutoencoderKL.from_pretrained(key, subfolder="vae").to(torch_device)
    tok = CLIPTokenizer.from_pretrained(key, subfolder="tokenizer")
    import os as _os
    from lgd_amd.clip import from_hf as _hip_text_encoder
    te = CLIPTextModel.from_pretrained(key, subfolder="text_encoder")
    if _os.environ.get("LGD_HF_TEXT", "0") == "1":       # keep the Hugging Face tower (A/B against the fp16-compute HIP one)
        te = te.to(torch_device)
    else:
        te = _hip_text_encoder(te, torch_device)         # HIP kernels: fp16 compute, 2-3e-3 of the fp32 tower's states

    class _HFVae:
        def decode(self, z):
            return vae.decode(z.to(vae.dtype)).sample
    if _os.environ.get("LGD_HF_VAE", "0") == "1":
        dec = _HFVae()
    else:
        from lgd_amd.vae import HipVAEDecoder
        dec = HipVAEDecoder.from_state_dict(vae.state_dict(), torch_device)      # models/models.py:41 on the HIP kernels
    return build_model_dict(cfg, {k: v.float() for k, v in hf.state_dict().items()}, vae=dec, tokenizer=tok,
                            text_encoder=te, scheduler_config=sched_cfg,
                            use_dpm_multistep_scheduler=use_dpm_multistep_scheduler)


def encode_prompts(tokenizer, text_encoder, prompts, negative_prompt="", return_full_only=False,
                   one_uncond_input_only=False):
    """models/models.py:63-89."""
    if negative_prompt == "":
        print("Note that negative_prompt is an empty string")
    text_input = tokenizer(prompts, padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                           return_tensors="pt")
    max_length = text_input.input_ids.shape[-1]
    n_unc = 1 if one_uncond_input_only else len(prompts)
    uncond_input = tokenizer([negative_prompt] * n_unc, padding="max_length", max_length=max_length, return_tensors="pt")
    with torch.no_grad():
        uncond_embeddings = text_encoder(uncond_input.input_ids.to(torch_device))[0]
        cond_embeddings = text_encoder(text_input.input_ids.to(torch_device))[0]
    if one_uncond_input_only:
        return uncond_embeddings, cond_embeddings
    text_embeddings = torch.cat([uncond_embeddings, cond_embeddings])
    if return_full_only:
        return text_embeddings
    return text_embeddings, uncond_embeddings, cond_embeddings


def process_input_embeddings(input_embeddings):
    """models/models.py:91-109: 2-tuple (uncond, cond) or 3-tuple (text, uncond, cond)."""
    assert isinstance(input_embeddings, (tuple, list))
    if len(input_embeddings) == 3:
        _, unc, cond = input_embeddings
        assert unc.shape[0] == cond.shape[0], f"{unc.shape[0]} != {cond.shape[0]}"
        return input_embeddings
    if len(input_embeddings) == 2:
        unc, cond = input_embeddings
        if unc.shape[0] == 1:
            unc = unc.expand(cond.shape)
        return torch.cat((unc, cond), dim=0), unc, cond
    raise ValueError(f"input_embeddings length: {len(input_embeddings)}")
