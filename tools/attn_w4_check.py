"""GPU-box tool: the d = 40 self-attention forward of attn_w4.hip (lgd_set_option("attn_w4", 2)) against fp32 torch
on ragged query / key counts, spiked keys (forces the reference raise) and large logits, then A/B timing against the
round-3 kernel (attn_w4 = 0) on the benchmark's shapes, interleaved rounds."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
d = 40


def run(B, H, S, Sk, mode, scale_q=1.0, spike=False, lse=False, seed=0):
    C = H * d
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(B, S, C, generator=g) * scale_q).to(dev).half()
    k = torch.randn(B, Sk, C, generator=g).to(dev).half()
    v = torch.randn(B, Sk, C, generator=g).to(dev).half()
    if spike:       # a few keys that dominate some queries late in the key sequence: the running reference must rise
        for j in (Sk // 2 + 3, Sk - 5):
            k[:, j] = q[:, (j * 7) % S] * 6.0
    o = torch.full((B, S, C), float("nan"), device=dev, dtype=torch.float16)
    L = torch.full((B, H, S), float("nan"), device=dev) if lse else None
    ops.set_option("attn_w4", mode)
    ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, d ** -0.5, lse=L)
    torch.cuda.synchronize()
    err = lerr = 0.0
    for b, h in ((0, 0), (B - 1, H - 1), (B // 2, H // 2)):
        sl = slice(h * d, (h + 1) * d)
        logits = q[b, :, sl].float() @ k[b, :, sl].float().t() * d ** -0.5
        ref = logits.softmax(-1) @ v[b, :, sl].float()
        err = max(err, float((o[b, :, sl].float() - ref).abs().max() / ref.abs().max()))
        if lse:
            lerr = max(lerr, float((L[b, h] - torch.logsumexp(logits, -1) * 1.4426950408889634).abs().max()))
    return err, lerr, bool(torch.isfinite(o.float()).all())


ok = True
for (B, H, S, Sk, kw) in [(2, 8, 4096, 4096, {}), (2, 8, 4096, 4126, {}), (1, 2, 300, 77, {}), (1, 3, 257, 64, {}), (2, 2, 64, 1, {}),
                          (1, 2, 1000, 129, dict(lse=True)), (2, 8, 4096, 4096, dict(spike=True, lse=True)),
                          (2, 4, 2048, 2111, dict(spike=True)), (2, 8, 4096, 4096, dict(scale_q=6.0, lse=True)),
                          (1, 1, 31, 200, {})]:
    e4, l4, f4 = run(B, H, S, Sk, 2, **kw)
    e0, l0, f0 = run(B, H, S, Sk, 0, **kw)
    bad = (e4 > 4e-3) or (l4 > 2e-2) or not f4
    ok &= not bad
    print(f"B{B} H{H} S{S}x{Sk} {kw}: w4 err {e4:.2e} lse {l4:.1e} finite {f4} | r3 err {e0:.2e} lse {l0:.1e}{'  WRONG' if bad else ''}", flush=True)
print("CORRECT" if ok else "MISMATCH")

if os.environ.get("TIME", "1") == "1":
    for (B, H, S, Sk) in [(16, 8, 4096, 4096), (8, 8, 4096, 4096), (16, 8, 4096, 4126), (4, 8, 4096, 4096)]:
        C = H * d
        g = torch.Generator().manual_seed(0)
        q = torch.randn(B, S, C, generator=g).to(dev).half()
        k = torch.randn(B, Sk, C, generator=g).to(dev).half()
        v = torch.randn(B, Sk, C, generator=g).to(dev).half()
        o = torch.empty(B, S, C, device=dev, dtype=torch.float16)
        times = {0: [], 2: []}
        for _ in range(6):
            for mode in (0, 2):
                ops.set_option("attn_w4", mode)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, d ** -0.5)
                e1.record(); torch.cuda.synchronize()
                times[mode].append(e0.elapsed_time(e1) * 100)
        line = f"time B{B} H{H} S{S}x{Sk}:"
        for mode, name in ((0, "round 3"), (2, "w4")):
            us = sorted(times[mode][1:])[2]
            line += f"  [{name}] {us:7.1f} us {4.0 * B * H * S * Sk * d / us / 1e6:6.1f} TF/s"
        print(line, flush=True)
ops.set_option("attn_w4", 1)
