"""generation/backward_guidance.py of the reference: plugin `backward_guidance` (layout-guidance
baseline: one guided generation, no per-box stage), on the HIP engine."""
import models
from lgd_amd.pipeline import backward_guidance_generate

from ._common import DEFAULT_OVERALL_NEGATIVE_PROMPT, DEFAULT_SO_NEGATIVE_PROMPT, EasyDict, build_layout, note_precision

version = "backward_guidance"
height = width = 512
num_inference_steps = 50            # module constants as at generation/backward_guidance.py:28-33
guidance_scale = 7.5


def run(spec, bg_seed=1, overall_loss_scale=30, overall_loss_threshold=0.2, overall_max_iter=5,
        overall_max_index_step=10):
    """generation/backward_guidance.py:43-137, defaults of :46-49.  The reference builds its guidance kwargs (:99-112)
    WITHOUT `use_ratio_based_loss`, so add_ca_loss_per_attn_map_to_loss runs its default, the ratio-based branch
    (utils/guidance.py:91,118-130): mean over heads of (1 - sum(A*M)/sum(A))^2 per phrase token; there are no
    reference maps (`ref_ca_saved_attns=None`), so `ref_ca_loss_weight=0.5` has no effect.  Pinned by
    tests/golden/run_backward_guidance_tiny.npz (the reference's own run(), oracle/make_golden_runs.py)."""
    # the reference runs this baseline without autocast, i.e. in the dtype models.load_sd loaded (fp32 by default)
    note_precision("generation.backward_guidance.run", "a run() without autocast")
    sm = models.model_dict.sampler
    lay = build_layout(spec, bg_seed, bg_seed, DEFAULT_SO_NEGATIVE_PROMPT, DEFAULT_OVERALL_NEGATIVE_PROMPT, height, width)
    out = backward_guidance_generate(sm, lay, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                                     loss_scale=overall_loss_scale, loss_threshold=overall_loss_threshold,
                                     max_iter=overall_max_iter, max_index_step=overall_max_index_step,
                                     height=height, width=width)
    return EasyDict(image=out["image"])
