"""ORACLE TEST INFRASTRUCTURE — runs the reference's OWN, unmodified Python on CPU.

Imports /root/reference/{models,utils} through oracle/stubs (diffusers 0.18.0 is not installed) with
three CPU monkeypatches:
  1. torch.Tensor.cuda -> identity            (utils/guidance.py:186,191,253,262,273)
  2. torch.zeros(device="cuda") -> cpu        (utils/guidance.py:104,204)
  3. utils.utils.torch_device = "cpu" BEFORE `import models` (by-value import at models/models.py:9,
     utils/latents.py:4, models/pipelines.py:9)
Only usable where /root/reference exists (the build container).  Never imported by the product.
"""
import os
import sys

import torch

REF_ROOT = os.environ.get("LGD_REFERENCE_ROOT", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
_ready = False


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def setup():
    """Idempotent: path + monkeypatches + import of the reference packages."""
    global _ready
    if _ready:
        return
    if not available():
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    for p in (os.path.join(_HERE, "stubs"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    if _REPO not in sys.path:
        sys.path.append(_REPO)
    torch.Tensor.cuda = lambda self, *a, **k: self
    _zeros = torch.zeros

    def zeros(*a, **k):
        if str(k.get("device", "")) .startswith("cuda"):
            k["device"] = "cpu"
        return _zeros(*a, **k)

    torch.zeros = zeros
    torch.cuda.empty_cache = lambda: None
    import utils.utils as ref_utils_utils  # noqa
    ref_utils_utils.torch_device = "cpu"
    import utils as ref_utils  # noqa
    ref_utils.torch_device = "cpu"
    import models  # noqa: F401  (star-imports models.models)
    _ready = True


def ref_modules():
    setup()
    import models
    import utils
    from models import pipelines, unet_2d_condition, attention_processor
    from utils import guidance, latents, schedule, attn
    return dict(models=models, utils=utils, pipelines=pipelines, unet_2d_condition=unet_2d_condition,
                attention_processor=attention_processor, guidance=guidance, latents=latents,
                schedule=schedule, attn=attn)


def build_ref_unet(cfg, seed=0, state_dict=None):
    """The reference's UNet2DConditionModel (models/unet_2d_condition.py:118) with the seeded
    synthetic weights of lgd_amd.weights (same fp32 tensors the HIP engine packs)."""
    setup()
    import lgd_amd  # noqa: F401
    from lgd_amd import weights
    from models.unet_2d_condition import UNet2DConditionModel
    unet = UNet2DConditionModel(**cfg.to_ref_kwargs())
    sd = state_dict if state_dict is not None else weights.synth_state_dict(cfg, seed)
    ref_sd = unet.state_dict()
    missing = set(ref_sd) - set(sd)
    extra = set(sd) - set(ref_sd)
    if missing or extra:
        raise RuntimeError(f"parameter inventory mismatch: missing {sorted(missing)[:5]} extra {sorted(extra)[:5]}")
    for k, v in ref_sd.items():
        if tuple(v.shape) != tuple(sd[k].shape):
            raise RuntimeError(f"shape mismatch for {k}: ref {tuple(v.shape)} vs {tuple(sd[k].shape)}")
    unet.load_state_dict(sd)
    unet.eval()
    return unet


class StubVAE(torch.nn.Module):
    """decode(latents) stand-in: pipelines.decode (pipelines.py:117-127) only needs `.decode(x).sample`."""

    class _Out:
        def __init__(self, s):
            self.sample = s

    def decode(self, z):
        return self._Out(torch.tanh(z[:, :3]))


def build_model_dict(cfg, seed=0, prediction_type=None):
    setup()
    from diffusers import DDIMScheduler
    from easydict import EasyDict
    unet = build_ref_unet(cfg, seed)
    sched = DDIMScheduler(prediction_type=prediction_type or cfg.prediction_type)
    return EasyDict(vae=StubVAE(), tokenizer=None, text_encoder=None, unet=unet, scheduler=sched,
                    dtype=torch.float32)
