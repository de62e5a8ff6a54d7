"""generation/boxdiff.py of the reference: plugin `boxdiff` (the BoxDiff baseline: one generation guided by
utils/boxdiff.py's energy, no per-box stage), on the HIP engine."""
import models
from lgd_amd.pipeline import boxdiff_generate
from lgd_amd.sampler import BOXDIFF_GUIDANCE_ATTN_KEYS

from ._common import DEFAULT_OVERALL_NEGATIVE_PROMPT, DEFAULT_SO_NEGATIVE_PROMPT, EasyDict, build_layout, note_precision

version = "boxdiff"
height = width = 512                # module constants as at generation/boxdiff.py:26-31
num_inference_steps = 50
guidance_scale = 7.5
overall_guidance_attn_keys = list(BOXDIFF_GUIDANCE_ATTN_KEYS)      # generation/boxdiff.py:33-39


def run(spec, bg_seed=1, overall_max_index_step=25):
    """generation/boxdiff.py:46-131: the overall prompt only (`parse.convert_spec`, negative prompt
    DEFAULT_OVERALL_NEGATIVE_PROMPT + the spec's extra one), noise from `bg_seed`, then
    generate_semantic_guidance(use_boxdiff=True) with `overall_guidance_attn_keys` and max_index_step =
    overall_max_index_step; `ref_ca_saved_attns=None` / `ref_ca_loss_weight=0.0` mean no reference-attention term.
    Pinned by tests/golden/run_boxdiff_tiny.npz (the reference's own run(), oracle/make_golden_boxdiff.py)."""
    # the reference runs this baseline without autocast, i.e. in the dtype models.load_sd loaded (fp32 by default)
    note_precision("generation.boxdiff.run", "a run() without autocast")
    sm = models.model_dict.sampler
    lay = build_layout(spec, bg_seed, bg_seed, DEFAULT_SO_NEGATIVE_PROMPT, DEFAULT_OVERALL_NEGATIVE_PROMPT, height, width)
    out = boxdiff_generate(sm, lay, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                           max_index_step=overall_max_index_step, height=height, width=width,
                           guidance_attn_keys=overall_guidance_attn_keys)
    return EasyDict(image=out["image"])
