import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import _lib
if os.environ.get('LGD_LIB'): _lib.LIB_PATH = os.environ['LGD_LIB']
from lgd_amd import ops
dev = torch.device("cuda:0")
def rnd(*shape, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dev)
def run(B, H, S, d, spikes, mult=6.0, force=None):
    if force is not None: ops.set_option("attn32", int(force))
    C = H * d
    q = rnd(B, S, C, seed=1).half(); k = rnd(B, S, C, seed=2).half(); v = rnd(B, S, C, seed=3).half()
    for qi, ki in spikes:
        k[:, ki] = q[:, qi] * mult
    o = torch.empty(B, S, C, device=dev, dtype=torch.float16)
    ops.attn_fwd(q, k, v, o, B, H, S, S, d, d ** -0.5)
    bad = ~torch.isfinite(o)
    nan, inf = int(torch.isnan(o).sum()), int(torch.isinf(o).sum())
    rows = sorted({(int(b), int(s), int(c) // d) for b, s, c in bad.nonzero().tolist()})
    sp = lambda t: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    info = ""
    if rows:
        b, s_, h = rows[0]
        sc = (sp(q)[b, h, s_] @ sp(k)[b, h].t()) * d ** -0.5 * 1.4427
        # running max per tile
        tm = sc.reshape(-1, 64).max(dim=1).values
        info = f" first bad row {rows[0]} argmax {int(sc.argmax())} max {float(sc.max()):.1f} tile maxes (first 4) {[round(float(x),1) for x in tm[:4]]} last 3 {[round(float(x),1) for x in tm[-3:]]}"
    print(f"B{B} S{S} d{d} spikes {spikes} x{mult} force={force}: bad rows {len(rows)} nan {nan} inf {inf}{info}", flush=True)
S = 4096
S = 1024
run(16, 8, S, 80, [(5, S - 3), (S // 2 + 1, S // 2 + 70), (S - 1, 200)], force="1")
run(16, 8, S, 80, [(5, S - 3)], force="1")
run(16, 8, S, 80, [(5, 200)], force="1")
run(16, 8, S, 80, [(5, 196)], force="1")
run(16, 8, S, 80, [(5, 70)], force="1")
run(16, 8, S, 80, [(5, 200)], mult=2.0, force="1")
run(16, 8, S, 80, [], force="1")
run(16, 8, S, 80, [(5, S - 3), (S // 2 + 1, S // 2 + 70), (S - 1, 200)], force="0")
