"""Shared host-side front end of the generation plugins: spec -> CachedLayout (text encoder + tokenizer
products), mirroring generation/lmd_plus.py:270-345,419-438 and utils/parse.py:311-367."""
import numpy as np
import torch

import models
from lgd_amd.pipeline import CachedLayout, convert_box
from utils import guidance

# prompt.py:43-44 of the reference (negative prompts are part of the plugin's default arguments)
DEFAULT_SO_NEGATIVE_PROMPT = "artifacts, blurry, smooth texture, bad quality, distortions, unrealistic, distorted image, bad proportions, duplicate, two, many, group, occlusion, occluded, side, border, collate"
DEFAULT_OVERALL_NEGATIVE_PROMPT = "artifacts, blurry, smooth texture, bad quality, distortions, unrealistic, distorted image, bad proportions, duplicate"


def _pluraliser():
    """The reference pluralises repeated object names with the third-party `inflect` package
    (utils/parse.py:7,11,343-346).  It is only needed for specs that repeat a name; without it such a spec is an
    error here rather than a silently different overall prompt (different text -> different tokens)."""
    try:
        import inflect
    except ImportError as e:
        raise RuntimeError("this spec repeats an object name; the overall prompt then needs the `inflect` package "
                           "(as the reference does) to pluralise it") from e
    return inflect.engine()


def convert_spec(spec, height, width, include_counts=True, verbose=False):
    """Spec -> (per-box (prompt, phrase, word, box) list, overall prompt, [(phrase, word, boxes)]) with the
    conventions of utils/parse.py:311-367: boxes sorted by object name (so that the per-box order equals the
    flattened overall order), one overall phrase per distinct name, repeated names counted and pluralised."""
    prompt, bg_prompt = spec['prompt'], spec['bg_prompt']
    named = sorted(((name, convert_box(box, height=height, width=width)) for name, box in spec['gen_boxes']),
                   key=lambda nb: nb[0])
    so = [((f"{bg_prompt} with {name}" if bg_prompt else f"{name}"), name, name.split(" ")[-1], box)
          for name, box in named]
    overall = []
    for name in sorted({n for n, _ in named}):                           # np.unique order
        boxes = [box for n, box in named if n == name]
        phrase = name
        if len(boxes) > 1:
            p = _pluraliser()
            phrase = p.plural_noun(name.replace("an ", "").replace("a ", ""))
            if include_counts:
                phrase = p.number_to_words(len(boxes)) + " " + phrase
        overall.append((phrase, phrase.split(' ')[-1], boxes))
    listed = ", ".join(ph for ph, _, _ in overall)
    overall_prompt = ((f"{bg_prompt} with {listed}" if bg_prompt else listed) if listed else bg_prompt)
    if verbose:
        print("so_prompt_phrase_word_box_list:", so, "overall_prompt:", overall_prompt)
    return so, overall_prompt, overall


def build_layout(spec, bg_seed, fg_seed_start, so_negative_prompt, overall_negative_prompt, height=512, width=512,
                 overall_prompt_override="", verbose=False):
    """Text side of lmd_plus.run / lmd.run for one spec -> CachedLayout (boxes are the ORIGINAL boxes; the
    optional centring of the per-box generations is done by the pipeline, lgd_amd.pipeline._centered_so_boxes)."""
    md = models.model_dict
    tok, te = md.tokenizer, md.text_encoder
    if tok is None or te is None:
        raise RuntimeError("model_dict has no tokenizer/text_encoder (needed to turn a spec into embeddings); "
                           "offline benchmarks use lgd_amd.pipeline.CachedLayout.synthetic instead")
    so_list, overall_prompt, overall = convert_spec(spec, height, width, verbose=verbose)
    if overall_prompt_override and overall_prompt_override.strip():
        overall_prompt = overall_prompt_override.strip()
    if spec.get("extra_neg_prompt"):
        so_negative_prompt = spec["extra_neg_prompt"] + ", " + so_negative_prompt
        overall_negative_prompt = spec["extra_neg_prompt"] + ", " + overall_negative_prompt
    n = len(so_list)
    cx = md.unet.config.cross_attention_dim
    if n:
        so_unc, so_cond = models.encode_prompts(prompts=[p for p, _, _, _ in so_list], tokenizer=tok, text_encoder=te,
                                                negative_prompt=so_negative_prompt, one_uncond_input_only=True)
    else:
        so_unc, so_cond = torch.zeros(1, 77, cx), torch.zeros(0, 77, cx)
    so_pos, so_word = [], []
    for p, ph, w, _ in so_list:
        pos, wi = guidance.get_phrase_indices(tok, p, [ph], words=[w], return_word_token_indices=True)
        so_pos.append(pos[0])
        so_word.append(wi[0])
    phrases, words = [o[0] for o in overall], [o[1] for o in overall]
    o_pos, o_word, overall_prompt = guidance.get_phrase_indices(tok, overall_prompt, phrases, words=words,
                                                                return_word_token_indices=True,
                                                                add_suffix_if_not_found=True)
    _, o_unc, o_cond = models.encode_prompts(prompts=[overall_prompt], tokenizer=tok, text_encoder=te,
                                             negative_prompt=overall_negative_prompt)
    # GLIGEN phrase embeddings = CLIP pooler_output of each box's phrase (pipelines.py:303-304)
    if n:
        ti = tok([ph for _, ph, _, _ in so_list], padding=True, return_tensors="pt").to("cuda")
        pe = te(**ti).pooler_output.float().cpu()
    else:
        pe = torch.zeros(0, 768)
    names = [ph for _, ph, _, _ in so_list]
    groups, k = [], 0
    for _, _, bbs in overall:
        groups.append(list(range(k, k + len(bbs))))
        k += len(bbs)
    return CachedLayout(boxes=[tuple(b) for _, _, _, b in so_list], so_uncond=so_unc.float().cpu(),
                        so_cond=so_cond.float().cpu(), so_object_positions=so_pos, so_word_token_index=so_word,
                        overall_uncond=o_unc.float().cpu(), overall_cond=o_cond.float().cpu(), overall_groups=groups,
                        overall_object_positions=o_pos, overall_word_token_indices=o_word, phrase_embeddings=pe,
                        bg_seed=bg_seed, fg_seed_start=fg_seed_start)


def sam_refiner(model_dict, height, width, **kw):
    """The SAM mask refiner of the plugin when `model_dict` carries a SAM model (generate.py:126-127 merges
    `sam.load_sam()` into it), else None = box masks."""
    has = ("sam_model" in model_dict) if isinstance(model_dict, dict) else hasattr(model_dict, "sam_model")
    if not has:
        return None
    from lgd_amd.sam_refine import SamRefiner
    md = model_dict if isinstance(model_dict, dict) else vars(model_dict)
    return SamRefiner(dict(sam_model=md["sam_model"], sam_processor=md["sam_processor"]), height=height, width=width, **kw)


_PRECISION_NOTED = set()


def note_precision(where: str, requested: str):
    """Precision contract of the drop-in: the HIP engine ALWAYS computes in fp16 with fp32
    accumulation, fp32 statistics / softmax / energy / latents — the arithmetic of the reference's autocast mode
    (generation/lmd_plus.py:226,336).  The reference runs training-free LMD and the layout-guidance baseline in fp32
    (`use_autocast=False`, generation/lmd.py:254,375; generation/backward_guidance.py has no autocast at all;
    `load_sd(use_fp16=False)`, models/models.py:16,33-38).  Such a request is honoured as "fp16 compute within the stated
    tolerance of the fp32 result" (tests/test_lmd_fullwidth_gpu.py), and said so ONCE per call site instead of silently."""
    if where in _PRECISION_NOTED:
        return False
    _PRECISION_NOTED.add(where)
    import warnings
    warnings.warn(f"{where}: {requested} asks for the reference's fp32 execution; the HIP engine computes in fp16 with fp32 "
                  "accumulation (fp32 latents, statistics, softmax and energy) — results match the fp32 reference within the "
                  "fp16 tolerances stated in INTEGRATION.md section 2, not bit for bit", RuntimeWarning, stacklevel=3)
    return True


class EasyDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
