"""Per-op comparison of the SAM port: every kernel call of lgd_amd/sam.py is also evaluated by the torch statement
(tests/ops_emul.py) on the same device inputs; prints the calls whose results differ (GPU box, debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, transformers
import lgd_amd, sam_cases, ops_emul
from lgd_amd import ops, sam as lsam

dev = torch.device("cuda:0")
TOL = float(os.environ.get("TOL", "2e-2"))


def relerr(a, b):
    a, b = a.float(), b.float()
    if not torch.isfinite(a).all():
        return float("nan")
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


class Both:
    n = 0

    def __getattr__(self, name):
        real, em = getattr(ops, name), getattr(ops_emul, name)
        if not callable(real):
            return real

        def f(*a, **k):
            Both.n += 1
            k_em = dict(k)
            a_em = list(a)
            if name == "attn_fwd":
                a_em[3] = torch.empty_like(a[3])
            if k.get("out") is not None:
                k_em["out"] = torch.empty_like(k["out"])
            r = real(*a, **k)
            torch.cuda.synchronize()
            e = em(*a_em, **k_em)
            rs, es = (r if isinstance(r, tuple) else (r,)), (e if isinstance(e, tuple) else (e,))
            errs = [relerr(x, y) for x, y in zip(rs, es)]
            bad = any(not (v < TOL) for v in errs)
            shapes = [tuple(x.shape) for x in a if torch.is_tensor(x)]
            if bad or os.environ.get("VERBOSE"):
                print(f"{'BAD ' if bad else 'ok  '}#{Both.n} {name} {shapes} strides {[x.stride() for x in a if torch.is_tensor(x)][:4]} "
                      f"kw {{{', '.join(f'{kk}={tuple(v.shape) if torch.is_tensor(v) else v}' for kk, v in k.items())}}} err {errs}", flush=True)
            return r
        return f


v = transformers.SamVisionConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, global_attn_indexes=[1], mlp_dim=128)
cfg = transformers.SamConfig(vision_config=v) if os.environ.get("SMALL_VISION", "1") == "1" else transformers.SamConfig()
hf = sam_cases.build_hf(transformers, cfg)
inp = sam_cases.inputs(cfg, B=1, P=2)
mine = lsam.HipSamModel(lsam.SamConfig.from_hf(cfg), hf.state_dict(), device=dev)
lsam.ops = Both()
got = mine(**inp)
print("calls", Both.n, "masks finite", bool(torch.isfinite(got.pred_masks).all()))
hf = hf.to(dev)
with torch.no_grad():
    want = hf(**{k: v.to(dev) for k, v in inp.items()})
print("masks", relerr(got.pred_masks, want.pred_masks), "iou", relerr(got.iou_scores, want.iou_scores))
