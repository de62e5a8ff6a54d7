"""GPU-box tool: enumerates every GEMM / implicit-conv launch shape of the engine's plans (SD1.4+GLIGEN,
64x64 latents: main passes of 1/2/4/8 images with the fuser on/off, guidance passes of 1/2/4 images fwd+bwd),
VERIFIES each (tile, split-K) candidate against a trusted tile's output, times the ones that pass with HIP events
and writes the winners to llm-groundeddiffusion_amd/tuning_gfx950.json.

    python tools/tune_gemm.py [config] [out.json]
"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops, weights, _lib
from lgd_amd.unet import UNetEngine
from lgd_amd.sampler import LMDSampler

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "sd14_gligen"
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "llm-groundeddiffusion_amd", "tuning_gfx950.json")
dev = torch.device("cuda:0")
cfg = weights.CONFIGS[cfg_name]
XL = cfg.addition_embed_type == "text_time"        # SDXL-refiner pass: shapes of one CFG UNet call + VAE encode / decode
eng = UNetEngine(cfg, dev, None, **(dict(max_text_batch=2) if XL else dict(max_text_batch=64)))      # zero weights are fine for timing shapes
eng.w.refresh_scalars()
L = cfg.sample_size

shapes = {}
orig = ops.gemm_launch
def rec(d, tag=None, flops=None):
    key = ops.shape_key(d)
    shapes.setdefault(key, dict(M=d.M, N=d.N, K=d.K, taps=d.taps, c0=d.c0, c1=d.c1, hin=d.hin, win=d.win,
                                hout=d.hout, wout=d.wout, stride=d.stride, ups=d.ups, epi=d.epi,
                                has_res=bool(d.res), has_bias=bool(d.bias), count=0))
    shapes[key]["count"] += 1
    return orig(d)
ops.gemm_launch = rec
if XL:
    from lgd_amd import vae
    eng.prepare_text(torch.zeros(2, 77, cfg.cross_attention_dim))
    eng.prepare_timesteps([281.0], dict(text_embeds=torch.zeros(2, cfg.pooled_dim), time_ids=torch.zeros(2, 5)))
    eng.set_step(0)
    plan = eng.plan(2, L)
    for _ in range(15):                      # 15 Euler steps per image against one encode + one decode
        plan.forward()
    vsd = vae.synth_aekl_state_dict()
    vae.HipVAEEncoder(vsd, dev).encode_moments(torch.zeros(1, 3, 8 * L, 8 * L))
    vae.HipVAEDecoder(vsd, dev).decode(torch.zeros(1, 4, L, L))
else:
    sm = LMDSampler(eng, use_graphs=False)
    eng.prepare_timesteps([500]); eng.set_step(0)
    batches = [int(x) for x in os.environ.get("LGD_TUNE_BATCHES", "1,2,4,8").split(",")]
    gbatches = [int(x) for x in os.environ["LGD_TUNE_GUIDE_BATCHES"].split(",") if x] if "LGD_TUNE_GUIDE_BATCHES" in os.environ \
        else [b for b in batches if b <= 4]
    for kind, fz, nb, fn in sm.profile_passes(L, 50, cfg.use_gated_attention, main_batches=batches,
                                              guide_batches=gbatches):
        fn()
if os.environ.get("LGD_TUNE_VAE") and not XL:
    # round 6: the VAE decodes inside the timed region (2.6 % of the default line) ran on heuristic tiles: their GEMM /
    # implicit-conv shapes for the decode batches the pipelines use (LGD_TUNE_VAE="1,2": one final image, two per-box images)
    from lgd_amd.vae import make_hip_vae
    hv = make_hip_vae(dev)
    n_before = len(shapes)
    for b_ in (int(x) for x in os.environ["LGD_TUNE_VAE"].split(",")):
        hv.decode(torch.zeros(b_, 4, L, L, device=dev))
    print(f"VAE decode: {len(shapes) - n_before} further shapes")
torch.cuda.synchronize()
ops.gemm_launch = orig
print(f"{len(shapes)} distinct GEMM shapes")

TILE_DIMS = {17: (128, 128), 18: (128, 64), 19: (64, 128), 20: (64, 64), 21: (32, 128), 22: (128, 160), 23: (64, 160),
             33: (256, 160), 34: (256, 128), 35: (256, 64), 37: (128, 160), 38: (128, 128),
             # round 4: the deeper-ringed small 8-wave tiles (5-6 LDS stages) were in the library but never candidates
             39: (128, 64), 40: (64, 160), 41: (64, 128), 42: (64, 64),
             # round 4: two-stage rings (plain single-source contractions only): 256x256, and 128x128 at two workgroups per CU
             44: (256, 256), 45: (128, 128),
             # round 6: phase-split 256-row tiles (plain single-source contractions and 3x3 stride-1 same-size convolutions)
             46: (256, 256), 47: (256, 320)}
PLAIN_ONLY = {44, 45}
PHASE = {46, 47}
NO_GEGLU = {22, 23, 33, 37, 40, 47}              # odd fragment counts cannot pair value | gate column blocks
VERIFY_TOL = 2e-3                                # fp16 outputs, different summation orders
rejected = []


def make_problem(sh):
    M, N, K = sh["M"], sh["N"], sh["K"]
    c0, c1, taps = sh["c0"], sh["c1"], sh["taps"]
    rows_in = M if taps == 1 else max(1, (M // max(sh["hout"] * sh["wout"], 1))) * sh["hin"] * sh["win"]
    g = torch.Generator(device=dev).manual_seed(M * 31 + N * 7 + K)
    a0 = torch.randn(rows_in, c0, generator=g, device=dev).half()
    a1 = torch.randn(rows_in, c1, generator=g, device=dev).half() if c1 else None
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).half()
    geglu = bool(sh["epi"] & 1)
    n_out = N // 2 if geglu else N
    bias = torch.randn(N, generator=g, device=dev) if sh["has_bias"] else None
    res = torch.randn(M, n_out, generator=g, device=dev).half() if sh["has_res"] else None
    return dict(a0=a0, a1=a1, w=w, bias=bias, res=res, n_out=n_out)


def make_desc(sh, pb, c, tile, splits):
    M, N, K = sh["M"], sh["N"], sh["K"]
    ws = torch.empty(splits * M * N, device=dev) if splits > 1 else None
    d = ops.gemm_desc(pb["a0"], pb["w"], c, M, N, K, a1=pb["a1"], c0=sh["c0"], c1=sh["c1"], lda0=sh["c0"],
                      lda1=sh["c1"], taps=sh["taps"], hin=sh["hin"], win=sh["win"], hout=sh["hout"], wout=sh["wout"],
                      stride=sh["stride"], ups=sh["ups"], bias=pb["bias"], res=pb["res"], ldr=pb["n_out"],
                      epi=sh["epi"] & 1, ldc=pb["n_out"], splits=splits, ws=ws, tile=tile)
    d._keep = ws
    return d


def reference(sh, pb):
    """Trusted output for this problem: the register-staged 64x128 main loop without split-K (tile code 3), which
    tests/test_ops_gpu.py checks against fp32 torch for every gather / epilogue variant."""
    c = torch.empty(sh["M"], pb["n_out"], device=dev, dtype=torch.float16)
    orig(make_desc(sh, pb, c, 3, 1))
    torch.cuda.synchronize()
    return c.float()


N_STREAMS = int(os.environ.get("LGD_TUNE_STREAMS", "1"))
STREAMS = [torch.cuda.Stream() for _ in range(N_STREAMS)] if N_STREAMS > 1 else []


def bench_concurrent(sh, pb, tile, splits, reps=6):
    """Cost of a launch when the GPU is shared by several launch sequences (lgd_amd/lanes.py): the same candidate on
    N_STREAMS streams at once, each with its own output and split-K scratch; us per launch = wall / (streams x reps).
    What counts here is how much of the machine a launch occupies for how long, not how soon it finishes alone."""
    ds = []
    for _ in STREAMS:
        c = torch.empty(sh["M"], pb["n_out"], device=dev, dtype=torch.float16)
        ds.append((make_desc(sh, pb, c, tile, splits), c))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    for st, (d, _) in zip(STREAMS, ds):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for _ in range(reps):
                orig(d)
    for st in STREAMS:
        cur.wait_stream(st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(STREAMS)) * 1e3


def bench(sh, pb, ref, tile, splits, reps=8):
    """Launches a candidate, VERIFIES its output against the trusted reference, then times it (us) —
    a candidate that computes something else is never recorded."""
    c = torch.zeros(sh["M"], pb["n_out"], device=dev, dtype=torch.float16)
    d = make_desc(sh, pb, c, tile, splits)
    try:
        orig(d)
    except RuntimeError:
        return None
    torch.cuda.synchronize()
    err = float((c.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    if not (err < VERIFY_TOL):
        rejected.append((ops.shape_key(d), tile, splits, err))
        return None
    if N_STREAMS > 1:
        return bench_concurrent(sh, pb, tile, splits)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        orig(d)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us

table = json.load(open(out_path)) if (os.path.exists(out_path) and not os.environ.get("LGD_TUNE_FRESH")) else {}
# LGD_TUNE_TOP=N: re-tune the N table entries that cost the passes the most time (count x us), keep the others
if os.environ.get("LGD_TUNE_TOP"):
    top = sorted((k for k in table if k in shapes), key=lambda k: -table[k]["us"] * shapes[k]["count"])[:int(os.environ["LGD_TUNE_TOP"])]
    covered = sum(table[k]["us"] * shapes[k]["count"] for k in top) / max(sum(table[k]["us"] * shapes[k]["count"] for k in table if k in shapes), 1e-9)
    print(f"re-tuning the {len(top)} most expensive shapes ({100 * covered:.1f} % of the passes' GEMM time)")
    old_entries = {k: table.pop(k) for k in top}
tot_old = tot_new = 0.0
for key, sh in sorted(shapes.items(), key=lambda kv: -kv[1]["count"] * kv[1]["M"] * kv[1]["N"] * kv[1]["K"]):
    if key in table or sh["K"] % 64:          # K % 64 != 0 (conv_in, K = 72): the register-staged fallback, heuristic tile
        continue
    M, N, K = sh["M"], sh["N"], sh["K"]
    geglu = bool(sh["epi"] & 1)
    pipe_ok = K % 64 == 0 and (sh["c0"] + sh["c1"]) % 64 == 0 and sh["c0"] % 64 == 0
    cands = []
    for tile, (bm, bn) in TILE_DIMS.items():
        if geglu and tile in NO_GEGLU:
            continue
        if tile > 32 and (not pipe_ok or M < bm):
            continue
        if tile in PLAIN_ONLY and (sh["taps"] != 1 or sh["c1"] > 0):
            continue
        if (tile in PLAIN_ONLY or tile in PHASE) and not key.endswith("_b1"):
            continue   # one matrix per launch: batched problems (the VAE's mid-block attention GEMMs) keep the older tiles
        if tile in PHASE and (sh["c1"] > 0 or (sh["taps"] == 9 and (sh["stride"] != 1 or sh["ups"] != 0 or sh["hin"] != sh["hout"]))):
            continue
        wgs = -(-M // bm) * -(-N // bn)
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):
            if sp > 1 and (K // 64 < 4 * sp or wgs * sp > 2048 or sp * M * N > (1 << 26)):
                continue
            if tile == 44 and sp > 1:
                continue   # the 256 x 256 tile leaves through the LDS epilogue only (one split)
            if wgs * sp < (8 if N_STREAMS > 1 else 48) and sp < 16 and K // 64 >= 8 * sp:
                continue   # hopelessly under-filled, a larger split exists (shared GPU: other sequences fill it)
            cands.append((tile, sp))
    only = os.environ.get("LGD_TUNE_ONLY_TILES")      # "44,45": try just these tiles against the entry the table holds
    if only:
        keep = {int(t) for t in only.split(",")}
        cands = [(t, sp) for t, sp in cands if t in keep and (sp == 1 or t != 44)]
        if not cands:
            if key in old_entries:
                table[key] = old_entries[key]
            continue
        if key in old_entries:
            cands.append((old_entries[key]["tile"], old_entries[key]["splits"]))
    pb = make_problem(sh)
    ref = reference(sh, pb)
    best = None
    from lgd_amd.unet import choose_splits
    base = bench(sh, pb, ref, ops.choose_tile(M, N, choose_splits(M, N, K), geglu, K), choose_splits(M, N, K)) or 0.0
    for tile, sp in cands:
        t = bench(sh, pb, ref, tile, sp)
        if t is not None and (best is None or t < best[0]):
            best = (t, tile, sp)
    fl = 2.0 * M * N * K
    table[key] = dict(tile=best[1], splits=best[2], us=round(best[0], 2), tflops=round(fl / best[0] / 1e6, 1),
                      base_us=round(base, 2), count=sh["count"])
    ref_us = base
    if N_STREAMS > 1:
        table[key]["streams"] = N_STREAMS
        if os.environ.get("LGD_TUNE_TOP") and key in old_entries:     # what the latency-tuned choice costs in this regime
            o = old_entries[key]
            t_old = bench(sh, pb, ref, o["tile"], o["splits"])
            if t_old is not None:
                ref_us = t_old
                table[key]["latency_choice"] = dict(tile=o["tile"], splits=o["splits"], us=round(t_old, 2))
                if t_old <= best[0] * 1.02:                            # keep the latency choice unless clearly beaten
                    best = (t_old, o["tile"], o["splits"])
                    table[key].update(tile=o["tile"], splits=o["splits"], us=round(t_old, 2), tflops=round(fl / t_old / 1e6, 1))
    tot_old += ref_us * sh["count"]; tot_new += best[0] * sh["count"]
    n_done = globals().get("n_done", 0) + 1
    globals()["n_done"] = n_done
    if n_done % 8 == 0:                                        # a killed run keeps what it measured
        merged = dict(old_entries, **table) if os.environ.get("LGD_TUNE_TOP") else table
        json.dump(merged, open(out_path, "w"), indent=0, sort_keys=True)
    print(f"{key:60s} n={sh['count']:3d} base {ref_us:7.1f}us -> tile {best[1]} split {best[2]:2d} {best[0]:7.1f}us "
          f"{fl / best[0] / 1e6:6.1f} TF/s", flush=True)
print(f"sum over passes: {tot_old/1e3:.2f} ms -> {tot_new/1e3:.2f} ms")
for r in rejected:
    print("REJECTED (wrong output):", r)
json.dump(table, open(out_path, "w"), indent=0, sort_keys=True)
print("wrote", out_path)
