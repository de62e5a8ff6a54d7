"""ORACLE TEST INFRASTRUCTURE — extracts the stage-1 layouts the reference's benchmark uses
(cache/cache_lmd_v0.1_gpt-4.json, cache/cache_demo_v0.1_gpt-4.json) by running the reference's OWN
parser (utils/parse.py: parse_input_with_negative + filter_boxes(scale_boxes=False), the
`--no-scale-boxes-default` path of generate.py:282-299) and writes them as a small data fixture:

    llm-groundeddiffusion_amd/data/layouts_lmd_v0.1_gpt-4.json   [{prompt, gen_boxes:[[name,[x,y,w,h]]..], bg_prompt, neg_prompt}]

Build container only (needs /root/reference).  The GPU box reads the fixture, never the reference.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402


def main():
    rh.setup()
    from utils import parse
    for name in ("lmd_v0.1_gpt-4", "demo_v0.1_gpt-4"):
        # the lmd_v0.1 layouts are the benchmark's workload (product data); the demo layouts are test inputs
        out_dir = os.path.join(ROOT, "llm-groundeddiffusion_amd", "data") if name.startswith("lmd_") else \
            os.path.join(ROOT, "tests", "golden")
        cache = json.load(open(os.path.join(rh.REF_ROOT, "cache", f"cache_{name}.json")))
        rows = []
        for prompt, responses in cache.items():
            for resp in responses:
                gen_boxes, bg_prompt, neg_prompt = parse.parse_input_with_negative(text=resp, no_input=True)
                gen_boxes = parse.filter_boxes(gen_boxes, scale_boxes=False)
                rows.append(dict(prompt=prompt, gen_boxes=[[n, list(b)] for n, b in gen_boxes],
                                 bg_prompt=bg_prompt, neg_prompt=neg_prompt))
        path = os.path.join(out_dir, f"layouts_{name}.json")
        json.dump(rows, open(path, "w"), separators=(",", ":"))
        hist = {}
        for r in rows:
            hist[len(r["gen_boxes"])] = hist.get(len(r["gen_boxes"]), 0) + 1
        print(name, len(rows), "layouts; boxes histogram", dict(sorted(hist.items())), "->", path)


if __name__ == "__main__":
    main()
