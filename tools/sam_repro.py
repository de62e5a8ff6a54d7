import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, transformers
import lgd_amd, sam_cases, ops_emul
from lgd_amd import ops, sam as lsam
dev = torch.device("cuda:0")
cap = {}
class Tap:
    def __getattr__(self, name):
        real = getattr(ops, name)
        if name != "attn_fwd":
            return real
        def f(*a, **k):
            r = real(*a, **k)
            torch.cuda.synchronize()
            if not torch.isfinite(a[3]).all() and "a" not in cap:
                cap["a"], cap["k"] = [x.clone() if torch.is_tensor(x) else x for x in a], k
            return r
        return f
v = transformers.SamVisionConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, global_attn_indexes=[1], mlp_dim=128)
cfg = transformers.SamConfig(vision_config=v)
hf = sam_cases.build_hf(transformers, cfg)
inp = sam_cases.inputs(cfg, B=1, P=2)
mine = lsam.HipSamModel(lsam.SamConfig.from_hf(cfg), hf.state_dict(), device=dev)
lsam.ops = Tap()
mine(**inp)
a, k = cap["a"], cap["k"]
q, kk, vv, o = a[:4]
B, H, Sq, Sk, d, scale = a[4:10]
print("shape", B, H, Sq, Sk, d, scale, k)
for n, t in (("q", q), ("k", kk), ("v", vv)):
    print(n, tuple(t.shape), "absmax", float(t.float().abs().max()), "finite", bool(torch.isfinite(t).all()))
bad = ~torch.isfinite(o)
print("non-finite outputs", int(bad.sum()), "of", o.numel(), "rows", bad.any(1).nonzero().flatten()[:20].tolist(), "cols", bad.any(0).nonzero().flatten()[:40].tolist())
o2 = torch.empty_like(o)
ops.attn_fwd(q, kk, vv, o2, B, H, Sq, Sk, d, scale, **k); torch.cuda.synchronize()
print("rerun on clones: non-finite", int((~torch.isfinite(o2)).sum()))
ref = torch.empty_like(o)
ops_emul.attn_fwd(q, kk, vv, ref, B, H, Sq, Sk, d, scale, **k)
ok = torch.isfinite(o2)
print("err on finite part", float((o2.float() - ref.float())[ok].abs().max()), "ref absmax", float(ref.float().abs().max()))
# logits range
qh = q.float().reshape(B, Sq, H, d).permute(0, 2, 1, 3); kh = kk.float().reshape(B, Sk, H, d).permute(0, 2, 1, 3)
s = scale * qh @ kh.transpose(-1, -2)
print("logit absmax", float(s.abs().max()), "max-min per row max", float((s.max(-1).values - s.min(-1).values).max()))
for mul in (0.25, 0.5):
    o3 = torch.empty_like(o)
    ops.attn_fwd((q.float() * mul).half(), kk, vv, o3, B, H, Sq, Sk, d, scale, **k); torch.cuda.synchronize()
    print("q x", mul, "non-finite", int((~torch.isfinite(o3)).sum()))
