"""Host side of the orchestration goldens (tests/golden/run_*.npz, made by oracle/make_golden_runs.py from the
reference's OWN generation/lmd_plus.run and generation/lmd.run): the spec -> prompts conversion and the
phrase -> token-position lookup of the drop-in plugins must reproduce every call the reference made, with the same
fake tokenizer."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lgd_amd  # noqa: E402,F401
from fake_text import FakeTokenizer  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _calls():
    out = []
    g = np.load(os.path.join(GOLD, "run_lmd_plus_tiny.npz"))
    out += json.loads(str(g["a_phrase_calls"])) + json.loads(str(g["b_phrase_calls"]))
    out += json.loads(str(np.load(os.path.join(GOLD, "run_lmd_tiny.npz"))["phrase_calls"]))
    g = np.load(os.path.join(GOLD, "run_backward_guidance_tiny.npz"))       # generation/backward_guidance.run
    out += json.loads(str(g["a_phrase_calls"])) + json.loads(str(g["b_phrase_calls"]))
    return out


def test_phrase_indices_match_every_reference_call():
    guidance = _load(os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin", "utils", "guidance.py"), "dropin_guidance")
    calls = _calls()
    assert len(calls) >= 11
    for c in calls:
        got = guidance.get_phrase_indices(FakeTokenizer(), c["prompt"], c["phrases"], words=c["words"] or None,
                                          return_word_token_indices=True, add_suffix_if_not_found=c["add_suffix"])
        assert json.loads(json.dumps(got)) == c["out"], c
    # a phrase that is absent from the prompt is appended behind "| " (guidance.py:33-36)
    pos, words, prompt = guidance.get_phrase_indices(FakeTokenizer(), "a photo of a table", ["a red cup"], words=["cup"],
                                                     return_word_token_indices=True, add_suffix_if_not_found=True)
    assert prompt == "a photo of a table| a red cup" and pos == [[6, 7, 8]] and words == [8]


def test_convert_spec_matches_the_prompts_the_reference_built(monkeypatch):
    """The prompts recorded from the reference's run() are what parse.convert_spec produced (with the `inflect`
    stand-in of oracle/stubs): the plugin front end must build the same strings, the same box order and grouping."""
    sys.modules.pop("inflect", None)
    monkeypatch.syspath_prepend(os.path.join(ROOT, "oracle", "stubs"))
    # the module under test imports `models` / `utils` of the plugin tree at import time: load only its function
    src = open(os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin", "generation", "_common.py")).read()
    from lgd_amd.pipeline import convert_box
    ns = dict(convert_box=convert_box)
    body = src[src.index("def _pluraliser():"):src.index("def build_layout(")]
    exec(compile(body, "_common_convert_spec", "exec"), ns)
    spec3 = dict(prompt="A photo of two apples on a table",
                 gen_boxes=[("an apple", [20, 120, 80, 80]), ("an apple", [140, 110, 90, 90]),
                            ("a wooden spoon", [60, 30, 120, 40])],
                 bg_prompt="A photo of a table", extra_neg_prompt="cartoon")
    so, overall_prompt, overall = ns["convert_spec"](spec3, 256, 256)
    g = np.load(os.path.join(GOLD, "run_lmd_plus_tiny.npz"))
    calls = json.loads(str(g["b_phrase_calls"]))
    assert [p for p, _, _, _ in so] == [c["prompt"] for c in calls[:3]]
    assert [[ph] for _, ph, _, _ in so] == [c["phrases"] for c in calls[:3]]
    assert overall_prompt == calls[3]["prompt"] and [o[0] for o in overall] == calls[3]["phrases"]
    assert [len(o[2]) for o in overall] == [1, 2]                     # "a wooden spoon" < "an apple" (sorted names)
    # without `inflect` a repeated name is an error, never a silently different prompt
    sys.modules.pop("inflect", None)
    monkeypatch.setattr(sys, "path", [p for p in sys.path if not p.endswith(os.path.join("oracle", "stubs"))])
    import builtins
    real_import = builtins.__import__

    def no_inflect(name, *a, **k):
        if name == "inflect":
            raise ImportError("inflect")
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", no_inflect)
    with pytest.raises(RuntimeError):
        ns["convert_spec"](spec3, 256, 256)
    so2, _, _ = ns["convert_spec"](dict(spec3, gen_boxes=spec3["gen_boxes"][1:]), 256, 256)   # no repeats: fine
    assert len(so2) == 2


def test_fp32_requests_are_answered_with_one_warning_per_call_site():
    """The reference runs LMD / the layout-guidance baseline in fp32 (generation/lmd.py:254,375; models/models.py:33-38);
    the HIP engine computes in fp16.  The drop-in says so ONCE per call site instead of
    silently accepting `use_autocast=False` / `use_fp16=False`."""
    import warnings
    src = open(os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin", "generation", "_common.py")).read()
    ns = {}
    body = src[src.index("_PRECISION_NOTED = set()"):src.index("class EasyDict(dict):")]
    exec(compile(body, "_common_note_precision", "exec"), ns)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert ns["note_precision"]("generation.lmd.run", "use_autocast=False") is True
        assert ns["note_precision"]("generation.lmd.run", "use_autocast=False") is False
        assert ns["note_precision"]("models.load_sd", "use_fp16=False") is True
    assert len(w) == 2 and all(issubclass(x.category, RuntimeWarning) for x in w)
    assert "fp16" in str(w[0].message) and "INTEGRATION.md" in str(w[0].message)
    # every plugin entry that accepts an fp32 request routes it here
    for mod, needle in (("lmd.py", 'note_precision("generation.lmd.run"'), ("lmd_plus.py", 'note_precision("generation.lmd_plus.run"'),
                        ("backward_guidance.py", 'note_precision("generation.backward_guidance.run"')):
        assert needle in open(os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin", "generation", mod)).read()
    assert 'note_precision("models.load_sd"' in open(os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin", "models", "models.py")).read()
