"""generation/gligen.py of the reference: plugin `gligen` (the plain GLIGEN baseline: ONE generation of the overall
prompt with every box of the layout as grounding input, no per-box stage, no attention guidance), on the HIP engine."""
import torch

import models
from models import pipelines
from utils import latents as latents_utils

from ._common import DEFAULT_OVERALL_NEGATIVE_PROMPT, EasyDict, convert_spec

version = "gligen"
height = width = 512
num_inference_steps = 50            # module constants as at generation/gligen.py:30-36
guidance_scale = 7.5
batch_size = 1


def run(spec, gligen_scheduled_sampling_beta=0.4, bg_seed=1):
    """generation/gligen.py:42-99: phrases / boxes in `parse.convert_spec`'s per-box order, the overall prompt against the
    overall negative prompt (with the spec's `extra_neg_prompt` in front), initial noise from `torch.manual_seed(bg_seed)`
    on the process-wide CPU generator (models.get_unscaled_latents) times the scheduler's `init_noise_sigma`, then
    `pipelines.generate_gligen` with its default arguments but the plugin's guidance scale and scheduled-sampling beta."""
    md = models.model_dict
    assert "gligen" in models.sd_key, models.sd_key               # generation/gligen.py:16
    so_list, prompt, _ = convert_spec(spec, height, width, verbose=False)
    phrases = [item[0] for item in so_list]                         # generation/gligen.py:56: the per-box PROMPTS, as the reference passes them
    bboxes = [item[-1] for item in so_list]
    negative_prompt = DEFAULT_OVERALL_NEGATIVE_PROMPT
    if spec.get("extra_neg_prompt"):
        negative_prompt = spec["extra_neg_prompt"] + ", " + negative_prompt
    input_embeddings = models.encode_prompts(prompts=[prompt], tokenizer=md.tokenizer, text_encoder=md.text_encoder,
                                             negative_prompt=negative_prompt)
    generator = torch.manual_seed(bg_seed)
    # the reference's model_dict.dtype is torch.float unless load_sd(use_fp16=True) (models/models.py:34-38; fp16 arithmetic comes
    # from autocast): the noise is drawn in fp32 — an fp16 draw consumes the generator differently — and stays fp32 (the sampler keeps fp32 latents)
    lat = latents_utils.get_unscaled_latents(batch_size, md.unet.config.in_channels, height, width, generator, torch.float32)
    lat = lat * md.scheduler.init_noise_sigma
    lat, images = pipelines.generate_gligen(md, lat, input_embeddings, num_inference_steps, bboxes, phrases,
                                            guidance_scale=guidance_scale,
                                            gligen_scheduled_sampling_beta=gligen_scheduled_sampling_beta)
    return EasyDict(image=images[0])
