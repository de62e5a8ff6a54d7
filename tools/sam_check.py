"""Prints the HIP-vs-Hugging-Face errors of the SAM port and times the image encoder (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, transformers
import lgd_amd, sam_cases
from lgd_amd import sam as lsam

dev = torch.device("cuda:0")
def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
CASES = (("small", sam_cases.small_config(transformers), 2, 2), ("vit_base", transformers.SamConfig(), 1, 2))
for name, cfg, B, P in [c for c in CASES if os.environ.get("ONLY", c[0]) == c[0]]:
    for points in ((False,) if os.environ.get("ONLY") else (False, True)):
        hf = sam_cases.build_hf(transformers, cfg)
        inp = sam_cases.inputs(cfg, B=B, P=P, points=points)
        mine = lsam.HipSamModel(lsam.SamConfig.from_hf(cfg), hf.state_dict(), device=dev)
        hf = hf.to(dev)
        with torch.no_grad():
            want = hf(**{k: v.to(dev) for k, v in inp.items()})
            want_emb = hf.get_image_embeddings(inp["pixel_values"].to(dev))
        got = mine(**inp)
        torch.cuda.synchronize()
        agree = float(((got.pred_masks > 0) == (want.pred_masks > 0)).float().mean())
        print(name, "points" if points else "boxes", "emb", relerr(mine.get_image_embeddings(inp["pixel_values"]), want_emb),
              "masks", relerr(got.pred_masks, want.pred_masks), "iou", relerr(got.iou_scores, want.iou_scores), "agree", agree,
              "| emb absmax", float(want_emb.abs().max()), "mask absmax", float(want.pred_masks.abs().max()), flush=True)
    px = inp["pixel_values"].to(dev)
    for _ in range(2):
        mine.encode_image(px)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5):
        mine.encode_image(px)
    torch.cuda.synchronize(); t_enc = (time.time() - t) / 5
    t = time.time()
    for _ in range(5):
        mine(**inp)
    torch.cuda.synchronize(); t_all = (time.time() - t) / 5
    with torch.no_grad():
        hf16 = hf.half()
        i16 = {k: v.to(dev).half() for k, v in inp.items()}
        for _ in range(2):
            hf16(**i16)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5):
            hf16(**i16)
        torch.cuda.synchronize(); t_hf = (time.time() - t) / 5
    print(name, f"encoder {t_enc*1e3:.2f} ms, full call {t_all*1e3:.2f} ms (B={B}, P={P}); HF fp16 eager on the same GPU {t_hf*1e3:.2f} ms", flush=True)
