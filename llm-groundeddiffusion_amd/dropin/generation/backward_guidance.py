"""generation/backward_guidance.py of the reference: plugin `backward_guidance` (layout-guidance
baseline: one guided generation, no per-box stage), on the HIP engine."""
import models
from lgd_amd.pipeline import backward_guidance_generate

from ._common import DEFAULT_OVERALL_NEGATIVE_PROMPT, DEFAULT_SO_NEGATIVE_PROMPT, EasyDict, build_layout

version = "backward_guidance"
height = width = 512
guidance_scale = 7.5


def run(spec, bg_seed=1, overall_loss_scale=30, overall_loss_threshold=0.2, overall_max_iter=5,
        overall_max_index_step=10):
    """generation/backward_guidance.py:46-49 defaults; the reference uses the ratio-free max-based loss
    with its default top-p/weights (fg/bg 0.2, weights 1.0)."""
    sm = models.model_dict.sampler
    lay = build_layout(spec, bg_seed, bg_seed, DEFAULT_SO_NEGATIVE_PROMPT, DEFAULT_OVERALL_NEGATIVE_PROMPT, height, width)
    out = backward_guidance_generate(sm, lay, num_inference_steps=50, guidance_scale=guidance_scale,
                                     loss_scale=overall_loss_scale, loss_threshold=overall_loss_threshold,
                                     max_iter=overall_max_iter, max_index_step=overall_max_index_step,
                                     height=height, width=width)
    return EasyDict(image=out["image"])
