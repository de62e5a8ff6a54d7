#!/usr/bin/env python
"""bench.py — images/sec of the LMD+ stage-2 hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload batch4|lmd_v0.1]

A "step" = one pass of the hot path over one batch of cached layouts on every rank.

  --workload batch4 (default; BASELINE config[1] "batch=4 cached layouts"): `--layouts` (4) two-box layouts
      of the lmd_v0.1 cache per rank per step.
  --workload sdxl_refiner (BASELINE config[4], generate.py --sdxl --sdxl-step-ratio 0.3): `--layouts` (4) images per rank
      per step through the SDXL-refiner img2img post-pass at 1024 x 1024 (generation/sdxl_refinement.py: VAE encode,
      int(50 x 0.3) = 15 CFG + Euler steps of the 2.26 B-parameter refiner UNet, VAE decode), seeded random weights of
      the real architectures, synthetic input images and text embeddings.
  --workload lmd_v0.1 (BASELINE config[3]): `--prompts` (100) layouts taken evenly from the 400-entry
      lmd_v0.1 cache (0..5 boxes each), partitioned over the ranks by cost (N+1 generations per layout,
      longest-processing-time first; the reference's manual version is generate.py:23-25,243-250), global
      prompt index preserved (seeds derive from it).  One step = every rank processes its whole share.

Every layout goes through the full LMD+ stage 2 (per-box GLIGEN generations, composition, overall generation
with cross-attention guidance, 50 DDIM steps each, VAE decodes; SAM replaced by box masks, text encoder outputs
synthetic = "cached layouts").  Weights: seeded random SD1.4+GLIGEN architecture (no checkpoints in the
sandbox).  One process per GPU: started by the driver under torch.distributed.run, or — when `--gpus N > 1`
is given without a rendezvous in the environment — re-launched by this script itself the same way.  Rank 0
builds the weights and RCCL-broadcasts the two weight arenas; there is no collective inside the step loop.

Prints ONE JSON line (rank 0) with the contract's fields plus `roofline` (dominant kernel; per-launch HIP
event timing) and, at N=1, `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROCPROF_SUMMARY = ("profiles/r06_bench_lanes1_kernel_stats.csv (one launch sequence alone: its per-launch averages are the "
                   "ones comparable with avg_launch_us); profiles/r06_bench_driver_kernel_stats.csv (the default command: "
                   "durations of kernels of different lanes overlap each other)")
MFMA_PEAK_F16 = 2.5e15       # dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12

# analytic 2*MAC TFLOP of one UNet call (sd14_gligen, 64x64 latents)
TF_MAIN_ON, TF_MAIN_OFF = 2.2736, 1.6065            # CFG forward B=2, GLIGEN fuser on / off
TF_GUIDE_ON, TF_GUIDE_OFF = 0.5872 + 0.6885, 0.4086 + 0.4585   # guidance fwd (to up.1.2) + dgrad to the latents
TF_SD21_MAIN, TF_SD21_GUIDE = 4.2982, 1.0579 + 1.311           # SD2.1 at 96x96 latents (BASELINE config 3)


def load_cache():
    """The 400 stage-1 layouts of the reference's lmd_v0.1 benchmark cache (parsed by the reference's own parser,
    oracle/make_layouts.py) — workload data of the package, not a test fixture."""
    return json.load(open(os.path.join(ROOT, "llm-groundeddiffusion_amd", "data", "layouts_lmd_v0.1_gpt-4.json")))


def algorithmic_tflop(n_boxes, n_steps, beta, iters_on, iters_off):
    """LMD+ image = (N+1) generations x (grounded + plain CFG calls) + guidance iterations (fuser on / off)."""
    n_on = int(beta * n_steps)
    return (n_boxes + 1) * (n_on * TF_MAIN_ON + (n_steps - n_on) * TF_MAIN_OFF) + iters_on * TF_GUIDE_ON + iters_off * TF_GUIDE_OFF


def layout_cost(n_boxes, n_steps=50, beta=0.4, iters_on=55, iters_off=10):
    """Cost of one layout for the rank partition = its algorithmic TFLOP: N + 1 generations, plus the
    guidance iterations of the overall stage — which exist only when the layout has boxes (25 % of the lmd_v0.1 cache
    has none: no per-box stage, no guidance; generation/lmd_plus.py:418-470).  With the default schedule an N >= 1
    layout costs (N + 1) x 93.7 + 78.8 TF, i.e. ~ N + 1.84 generations, an empty one 1."""
    if n_boxes == 0:
        iters_on = iters_off = 0
    return algorithmic_tflop(n_boxes, n_steps, beta, iters_on, iters_off)


def partition_by_cost(costs, world):
    """Longest-processing-time-first assignment of items (cost_i) to `world` ranks; deterministic.
    Returns per-rank lists of item indices (ascending)."""
    load = [0.0] * world
    mine = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        load[r] += costs[i]
        mine[r].append(i)
    return [sorted(m) for m in mine]


def lanes_for(workload, n_layouts, max_lanes, min_layouts_per_lane=8):
    """Pipelines per GPU, derived from the work a rank actually has.  The per-step workloads hand every lane whole steps
    (always enough); the prompt-set workload splits the rank's layouts over the lanes, and a lane with only a few layouts
    runs small, badly filled UNet calls (100 prompts over 8 ranks = 12-13 layouts per rank: four lanes would get 3 each) —
    so a lane is only added per `min_layouts_per_lane` layouts."""
    if workload != "lmd_v0.1":
        return max(1, max_lanes)
    return max(1, min(max_lanes, n_layouts // max(1, min_layouts_per_lane)))


def padded_work(box_counts, max_batch, max_batch_guided, buckets, n_steps=50):
    """(real, padded) algorithmic TFLOP of ONE LMD+ denoising job over layouts with these box counts, as
    LMDSampler.denoise_batch packs it (sampler.plan_chunks): stage A = one unguided generation per box, stage B = one
    guided generation per layout with boxes + one unguided per layout without.  An inert copy that pads a call costs what
    a real image of that call costs (it rides through the guidance passes of a guided call as well)."""
    from lgd_amd.sampler import plan_chunks
    gen = layout_cost(0, n_steps)                      # one unguided generation
    gen_g = layout_cost(1, n_steps) - gen              # one guided generation (the overall stage of a layout with boxes)
    real = pad = 0.0
    for n, cap, cost in ((sum(box_counts), max_batch, gen), (sum(1 for b in box_counts if b), max_batch_guided, gen_g),
                         (sum(1 for b in box_counts if not b), max_batch, gen)):
        for count, bucket in plan_chunks(n, cap, buckets):
            real += count * cost
            pad += (bucket - count) * cost
    return real, pad


def select_prompts(rows, n):
    """`n` cache entries spread evenly over the file (its four prompt categories are stored in blocks)."""
    n = min(n, len(rows))
    return [(i * len(rows)) // n for i in range(n)]


def cpu_baseline(cfg, generations, n_steps, beta, iters_on, iters_off):
    """The CPU oracle (oracle/restate.py: fp32 restatement of the reference path, pinned against the
    reference's own code) timed on this box's host cores on a bounded sample (~20-30 s): one CFG UNet call with
    the GLIGEN fuser on, one with it off, one guidance iteration (fwd+bwd); extrapolated to a full LMD+ image
    with the call / iteration counts of the GPU run.  Threads are capped at 32: the oracle's fp32 convolutions
    get slower, not faster, when oversubscribed across a 256-thread host (measured 155 s vs 8 s per call)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restate as R
    from lgd_amd import weights
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = weights.synth_state_dict(cfg, 0)
    cd = dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
              attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups,
              norm_eps=cfg.norm_eps, gligen_positive_len=cfg.gligen_positive_len)
    L = cfg.sample_size
    x = torch.randn(2, 4, L, L)
    unc, cond = weights.synth_embeddings(cfg, 1)
    ehs = torch.cat([unc, cond])
    boxes = [[0.15, 0.35, 0.5, 0.8], [0.6, 0.38, 0.98, 0.8]]
    gl = None
    if cfg.use_gated_attention:
        b, e, m, _ = R.prepare_gligen_condition([boxes], [torch.randn(2, cfg.gligen_positive_len)])
        gl = dict(boxes=b, positive_embeddings=e, masks=m)
    times = {}
    with torch.no_grad():
        for on in ([True, False] if gl is not None else [False]):
            t0 = time.time()
            R.unet_forward(sd, cd, x, 500, ehs, gligen=gl, fuser_enabled=on)
            times[on] = time.time() - t0
    t_on, t_off = times.get(True, times[False]), times[False]
    sched = R.DDIM(prediction_type=cfg.prediction_type)
    sched.set_timesteps(n_steps)
    tg = {}
    for on in ([True, False] if gl is not None else [False]):      # guidance iterations with the fuser on AND off
        t0 = time.time()
        R.latent_backward_guidance(sd, cd, sched, cond, 0, boxes, [[1, 2, 3], [5, 6, 7]], sched.timesteps[0], x[:1],
                                   torch.tensor(1e4), loss_scale=5, loss_threshold=0.0, max_iter=1, max_index_step=30,
                                   guidance_attn_keys=R.DEFAULT_GUIDANCE_ATTN_KEYS, use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2,
                                   fg_weight=1.0, bg_weight=4.0, fuser_enabled=on,
                                   gligen=dict(boxes=gl["boxes"][:1], positive_embeddings=gl["positive_embeddings"][:1],
                                               masks=gl["masks"][:1]) if gl else None)
        tg[on] = time.time() - t0
    tg_on, tg_off = tg.get(True, tg[False]), tg[False]
    n_on = int(beta * n_steps) if gl is not None else 0
    per_image = generations * (n_on * t_on + (n_steps - n_on) * t_off) + iters_on * tg_on + iters_off * tg_off
    return dict(value=1.0 / per_image, unit="images/s", cores=cores, kind="port",
                sample=(f"oracle/restate.py fp32 on {cores} host threads, measured: 1 CFG UNet call (B=2) fuser on "
                        f"{t_on:.2f}s, fuser off {t_off:.2f}s, 1 guidance iteration (fwd+bwd, early exit) fuser on "
                        f"{tg_on:.2f}s, fuser off {tg_off:.2f}s; extrapolated to one image = {generations:.2f} generation(s) x "
                        f"({n_on} on + {n_steps - n_on} off) UNet calls + {iters_on:.1f} + {iters_off:.1f} guidance "
                        f"iterations (fuser on + off; VAE excluded)"))


def d40_attention_bounds():
    """What the SIMDs allow the d = 40 self-attention forward (csrc/attn_w4.hip), from MEASURED instruction costs and
    counts: per 32-query x 32-key score block a wave issues 7 `v_mfma_f32_32x32x16_f16` (3 for Q K^T over d padded to 48,
    4 for P V over d padded to 64) and 47.8 VALU wave-instructions, 16 of them `v_exp_f32` (one per 64 scores) — PMC of
    the B = 16, S = 4096 launch: SQ_INSTS_MFMA 14.68 M, SQ_INSTS_VALU 100.26 M (profiles/r04_attn_d40_w4_kernel_pmc_summary.json);
    cost per wave-instruction and SIMD at one wave per SIMD: MFMA 18.5 ns, v_exp_f32 5.04 ns, plain VALU 1.43 ns
    (profiles/r03_ubench.txt, at the clock the chip sustains).  MFMA and VALU streams of a SIMD measured additive there
    ("series"); "overlapped" is the bound if the shorter stream hid completely behind the longer one.  Algorithmic flops
    of a block: 4 x 32 x 32 x 40."""
    t_mfma = 7 * 18.5e-9
    t_valu = 16 * 5.04e-9 + (47.8 - 16) * 1.43e-9
    fl = 4.0 * 32 * 32 * 40
    n_simd = 256 * 4
    return dict(series_tflops=round(fl / (t_mfma + t_valu) * n_simd / 1e12, 1),
                overlapped_tflops=round(fl / max(t_mfma, t_valu) * n_simd / 1e12, 1),
                mfma_only_tflops=round(fl / t_mfma * n_simd / 1e12, 1),
                mfma_ns_per_block=round(t_mfma * 1e9, 1), valu_ns_per_block=round(t_valu * 1e9, 1),
                source="profiles/r04_attn_d40_w4_kernel_pmc_summary.json (instruction counts), profiles/r03_ubench.txt "
                       "(ns per wave-instruction per SIMD); d padded 40 -> 48 / 64 costs 29 % of the MFMA slots")


class ClockSampler:
    """Shader clock of this rank's GPU while the timed region runs: the current level of the driver's
    `pp_dpm_sclk` table (sysfs; a file read every 0.25 s from a host thread that otherwise sleeps), so that a line's
    TF/s can be read against the clock the chip actually held (the 2.5 PF peak is quoted at 2.4 GHz).  The file lists
    "0: min, 1: CURRENT *, 2: max" on this driver (fine-grained DPM).  Reports None where the file is absent, the device's
    PCI address cannot be matched or there is no current-level mark."""

    def __init__(self, index: int, period: float = 0.25):
        import glob
        import threading
        # the box may expose the sysfs nodes of every GPU of the node: take the card whose PCI address is THIS device's
        self.path = None
        try:
            import torch
            p = torch.cuda.get_device_properties(index)
            want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}."
            for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
                if want in os.path.realpath(os.path.dirname(f)):
                    self.path = f
        except Exception:          # noqa: BLE001 - no PCI identity, no clock record
            pass
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._thread = threading.Thread(target=self._run, name="lgd-clock-sampler", daemon=True)

    def read(self):
        try:
            for line in open(self.path):
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().lower().replace("*", "").replace("mhz", "").strip())
        except Exception:          # noqa: BLE001 - a missing / unreadable file means "not recorded"
            pass
        return None

    def _run(self):
        while not self._stop.wait(self.period):
            v = self.read()
            if v is not None:
                self.samples.append(v)

    def __enter__(self):
        if self.path:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.path:
            self._thread.join(timeout=2)

    def summary(self):
        if not self.samples:
            return None
        xs = self.samples
        return dict(mean_mhz=round(sum(xs) / len(xs), 1), min_mhz=min(xs), max_mhz=max(xs), samples=len(xs),
                    source=f"{self.path}, current level every {self.period}s over the timed region")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn(n):
    """`python bench.py --gpus N` without a rendezvous in the environment: start N ranks of this script under
    torch.distributed.run (one process per GPU), exactly as the driver would."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main_sdxl(args):
    """BASELINE config[4]: the SDXL-refiner post-pass (generation/sdxl_refinement.py), `--layouts` images per rank per
    step; the same contract line.  Images shard over ranks and lanes with no collective in the loop."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import lgd_amd  # noqa: F401
    from lgd_amd import dist as ldist, ops, sdxl, vae, weights
    from lgd_amd.lanes import LanePool, make_lanes
    from lgd_amd.unet import UNetEngine
    cfg = weights.CONFIGS[args.config]
    pinned = ldist.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), lanes=max(1, args.lanes))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        ldist.init(backend="nccl")
    eng = UNetEngine(cfg, dev, weights.synth_state_dict(cfg, 0) if rank == 0 else None, max_text_batch=2)
    bcast_s = ldist.broadcast_weights(eng.w, src=0) if world > 1 else 0.0
    tiny = cfg.block_out_channels[0] < 320
    vsd = vae.synth_aekl_state_dict((64, 128, 128, 128) if tiny else (128, 256, 512, 512), 1 if tiny else 2, seed=0)
    side = 8 * cfg.sample_size                                          # 1024 for the refiner
    T, ratio, gs = args.num_inference_steps, args.sdxl_step_ratio, 5.0

    def make(e):
        return sdxl.SDXLRefiner(e, vae.HipVAEEncoder(vsd, dev), vae.HipVAEDecoder(vsd, dev))
    lanes = make_lanes(eng, max(1, args.lanes), make)
    pool = LanePool(lanes, device=dev)
    g = torch.Generator().manual_seed(1000 + rank)
    n_img = args.layouts
    images = [(torch.rand((1, 3, side, side), generator=g) * 2 - 1).to(dev) for _ in range(n_img)]
    embeds = [(torch.randn((2, 77, cfg.cross_attention_dim), generator=g), torch.randn((2, cfg.pooled_dim), generator=g))
              for _ in range(n_img)]

    def one(lane, i):
        return lane.sampler.refine(images[i], embeds[i][0], embeds[i][1], seed=rank * n_img + i, strength=ratio,
                                   num_inference_steps=T, guidance_scale=gs, output="float")
    t_pre = time.perf_counter()
    for k in range(len(lanes)):                                         # plans + graph of every lane, one by one
        pool.map(one, [0], pin=[k])
    torch.cuda.synchronize()
    prebuild_s = time.perf_counter() - t_pre
    for _ in range(args.warmup):
        pool.map(one, list(range(n_img)))
    ldist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = pool.map(one, list(range(n_img)))
    torch.cuda.synchronize()
    busy = time.perf_counter() - t0
    ldist.barrier()
    dt = ldist.max_over_ranks(time.perf_counter() - t0)
    per_rank_busy = ldist.gather_floats(busy)
    pool.close()
    ldist.shutdown()
    if rank != 0:
        return
    assert all(bool(torch.isfinite(o).all()) for o in outs)
    n_images = args.steps * n_img * world
    n_run = T - lanes[0].sampler.scheduler.img2img_start(T, ratio)
    # algorithmic work and the dominant kernel: ONE image eagerly (no graph) on lane 0 with HIP events around every launch
    roofline, tf = None, None
    if not args.no_roofline:
        r0 = lanes[0].sampler
        r0.use_graphs = False
        r0.refine(images[0], *embeds[0], seed=0, strength=ratio, num_inference_steps=T, output="float")
        torch.cuda.synchronize()
        prof = ops.LaunchProfiler(max_records=200000)
        ops.PROFILER = prof
        r0.refine(images[0], *embeds[0], seed=0, strength=ratio, num_inference_steps=T, output="float")
        ops.PROFILER = None
        agg = prof.summary()
        tf = sum(v["flops"] for v in agg.values()) / 1e12
        name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
        ach = a["flops"] / (a["ms"] * 1e-3)
        gem = [v for k, v in agg.items() if k.startswith("gemm")]
        roofline = dict(bound="mfma", kernel=name, achieved=round(ach / 1e12, 2), peak=MFMA_PEAK_F16 / 1e12, unit="TFLOP/s",
                        frac=round(ach / MFMA_PEAK_F16, 4), traffic=None, avg_launch_us=round(a["ms"] * 1e3 / a["n"], 2),
                        launches_per_image=int(a["n"]), est_ms_per_image=round(a["ms"], 1),
                        traffic_note="no PMC pass for this workload", traffic_source=None,
                        method="HIP events around each launch of ONE image refined eagerly right after the timed region "
                               "(the timed region replays one hipGraph per step on config.lanes_per_gpu streams)",
                        all_gemm_tflops=round(sum(v["flops"] for v in gem) / max(sum(v["ms"] for v in gem) * 1e-3, 1e-12) / 1e12, 1),
                        all_kernels={k: dict(ms_per_image=round(v["ms"], 1), launches_per_image=int(v["n"]),
                                             tflops=round(v["flops"] / max(v["ms"] * 1e-3, 1e-12) / 1e12, 1))
                                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:12]})
    res = dict(metric=f"images/sec (SDXL refiner img2img, step ratio {ratio}, {side}^2)", value=round(n_images / dt, 4),
               unit="images/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt * 1e3 / args.steps, 1),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="fp16", data="synthetic",
               config=dict(workload=f"SDXL-refiner post-pass (generation/sdxl_refinement.py), {n_img} images/GPU/step, "
                                    f"{side}x{side}, {n_run} of {T} Euler steps (ratio {ratio}), guidance {gs}, {args.config} "
                                    f"({weights.num_params(cfg) / 1e9:.2f} B-parameter UNet, SD/SDXL VAE encoder + decoder, "
                                    "seeded random weights), VAE encode and decode included",
                           images_per_gpu=n_img, num_inference_steps=T, steps_run=n_run, parallelism=f"dp{world}",
                           rccl_ranks=world, lanes_per_gpu=len(lanes), gemm_tuning=lanes[0].engine.tuning_mode or ops.current_tuning_mode(),
                           algorithmic_tflop_per_image=round(tf, 2) if tf else None,
                           weight_broadcast_s=round(bcast_s, 3), prebuild_s=round(prebuild_s, 2),
                           per_rank_busy_s=[round(b, 3) for b in per_rank_busy],
                           per_rank_idle_s=[round(max(dt - b, 0.0), 3) for b in per_rank_busy],
                           host=dict(torch_threads=1, rank0_cpu_affinity=(f"{pinned[0]}-{pinned[-1]}" if pinned else "unpinned"))),
               roofline=roofline)
    if tf:
        res["config"]["whole_path_frac_of_mfma_peak"] = round(tf * 1e12 * (n_images / dt) / world / MFMA_PEAK_F16, 4)
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_sdxl(cfg, vsd, side, n_run, tf)
    print(json.dumps(res))


def cpu_baseline_sdxl(cfg, vsd, side, n_run, tflop_per_image):
    """oracle/restate_sdxl.py (fp32, host threads) on a bounded sample: ONE UNet call of the CFG pair at a reduced
    latent size, scaled to the full pass by algorithmic work."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import restate_sdxl as X
        from lgd_amd import weights
        n_thr = min(32, os.cpu_count() or 1)
        torch.set_num_threads(n_thr)
        sd = weights.synth_state_dict(cfg, 0)
        Ls = min(cfg.sample_size, 32)
        x = torch.randn(2, 4, Ls, Ls)
        ehs, pooled = torch.randn(2, 77, cfg.cross_attention_dim), torch.randn(2, cfg.pooled_dim)
        t0 = time.perf_counter()
        with torch.no_grad():
            X.unet_forward_xl(sd, cfg, x, 281.0, ehs, dict(text_embeds=pooled, time_ids=X.add_time_ids(side, side)))
        dt = time.perf_counter() - t0
        scale = (cfg.sample_size / Ls) ** 2                     # convolutions / projections; attention grows faster
        per_image = dt * scale * n_run
        return dict(value=1.0 / per_image, unit="images/s", cores=n_thr, kind="port",
                    sample=f"oracle/restate_sdxl.py fp32 on {n_thr} host threads: one CFG-pair UNet call at {Ls}x{Ls} latents "
                           f"{dt:.1f}s, scaled by pixel count to {cfg.sample_size}x{cfg.sample_size} x {n_run} steps (lower bound: "
                           "self-attention grows quadratically; VAE excluded)")
    except Exception as e:   # reported, never gating
        return dict(value=None, unit="images/s", cores=os.cpu_count(), kind="port", sample=f"failed: {e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="batch4", choices=["batch4", "lmd", "lmd_v0.1", "backward_guidance", "sdxl_refiner"])
    ap.add_argument("--sdxl-step-ratio", type=float, default=0.3, help="sdxl_refiner: img2img strength (generate.py:52)")
    ap.add_argument("--layouts", type=int, default=4, help="batch4: cached layouts per rank per step")
    ap.add_argument("--prompts", type=int, default=100, help="lmd_v0.1: prompts of the cache (whole job)")
    ap.add_argument("--config", default=None, help="default: sd14_gligen (sd21 for --workload backward_guidance)")
    ap.add_argument("--num-inference-steps", type=int, default=50)
    ap.add_argument("--lanes", type=int, default=4,
                    help="concurrent denoising pipelines per GPU, each on its own HIP stream with its own engine state "
                         "(lgd_amd/lanes.py): steps (batch4) or halves of the prompt set (lmd_v0.1) run side by side")
    ap.add_argument("--group", type=int, default=1,
                    help="batch4 / lmd: steps a lane takes AT ONCE (their layouts share UNet calls: --group 2 with --layouts 4 "
                         "= 16 per-box generations and 8 overall generations per denoising call, chunked by --max-batch*); "
                         "0 = auto: the steps split evenly over the lanes, at most --group-max per job")
    ap.add_argument("--group-max", type=int, default=5, help="--group 0: most steps a lane job may take")
    ap.add_argument("--max-batch", type=int, default=8, help="images per UNet call, unguided generations (LMDSampler.max_batch)")
    ap.add_argument("--max-batch-guided", type=int, default=8,
                    help="images per UNet call, guided generations (the default workload has 4 per step; the guided per-box "
                         "stage of --workload lmd has 8: one call of 8 measured +5.8 %% over two of 4)")
    ap.add_argument("--min-layouts-per-lane", type=int, default=8,
                    help="lmd_v0.1: a lane is added per this many layouts of the rank's share (at most --lanes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--shape-profile", default=None, metavar="PATH",
                    help="also write the per-launch-shape table of the roofline replay (ms / launches / TF/s per image) to PATH")
    ap.add_argument("--sam", action="store_true",
                    help="refine every per-box mask with SAM (sam-vit-base architecture, seeded random weights, "
                         "device-side processor) as real runs do; the default uses box masks (SURVEY.md 8d)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the rank(s) under torch.distributed.run and initialise RCCL even for --gpus 1: the rendezvous, "
                         "rank pinning, weight broadcast, barrier and gather code path of an N-GPU run, rehearsed on one GPU")
    ap.add_argument("--cpu-dryrun", action="store_true",
                    help="rendezvous / weight broadcast / partition / timing collectives on CPU (gloo), no GPU work")
    args = ap.parse_args()

    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn(args.gpus))
    if args.config is None:
        args.config = {"backward_guidance": "sd21", "sdxl_refiner": "sdxl_refiner", "lmd": "sd15"}.get(args.workload, "sd14_gligen")
    if args.workload == "sdxl_refiner":
        return main_sdxl(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import lgd_amd  # noqa: F401
    from lgd_amd import dist as ldist, weights
    from lgd_amd.pipeline import CachedLayout
    cache = load_cache()
    cfg = weights.CONFIGS["tiny_gligen" if args.cpu_dryrun else args.config]
    T = args.num_inference_steps
    beta = 0.4                                                      # lmd_plus.py:208-209

    # ---- this rank's share of the work (global prompt index preserved: seeds derive from it) -----------------
    if args.workload in ("batch4", "lmd", "backward_guidance"):
        pool = [i for i, r in enumerate(cache) if len(r["gen_boxes"]) == 2]
        mine = [pool[(rank * args.layouts + i) % len(pool)] for i in range(args.layouts)]
        seeds = [rank * args.layouts + i for i in range(args.layouts)]
        n_total = args.layouts * world
    else:
        sel = select_prompts(cache, args.prompts)
        parts = partition_by_cost([layout_cost(len(cache[i]["gen_boxes"]), args.num_inference_steps) for i in sel], world)
        mine = [sel[j] for j in parts[rank]]
        seeds = list(mine)
        n_total = len(sel)
    lays = [CachedLayout.synthetic(cfg, [(n, b) for n, b in cache[i]["gen_boxes"]], index=s) for i, s in zip(mine, seeds)]
    my_boxes = sum(l.n_boxes for l in lays)

    if args.cpu_dryrun:
        from lgd_amd.weightstore import WeightStore
        pinned = ldist.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), lanes=max(1, args.lanes))
        if world > 1:
            ldist.init(backend="gloo")
        ws = WeightStore(cfg, "cpu")
        if rank == 0:
            ws.load_state_dict(weights.synth_state_dict(cfg, 0))
        bcast_s = ldist.broadcast_weights(ws, src=0, chunk_bytes=8 << 20)
        ldist.barrier()
        t0 = time.perf_counter()
        # the rank's lanes as host threads: the same hand-out as the GPU path (whole steps, or one cost-balanced share
        # of the rank's layouts per lane), each job "running" for a time proportional to its algorithmic cost
        from lgd_amd.lanes import Lane, LanePool
        from lgd_amd.sampler import LMDSampler
        costs = [layout_cost(l.n_boxes, args.num_inference_steps) for l in lays]
        my_cost = sum(costs)
        n_lanes = lanes_for(args.workload, len(lays), max(1, args.lanes), args.min_layouts_per_lane)
        if args.workload == "lmd_v0.1" and n_lanes > 1:
            shares = partition_by_cost(costs, n_lanes)
        else:
            shares = [list(range(len(lays)))]
        jobs = [[costs[j] for j in sh] for sh in shares] if args.workload == "lmd_v0.1" else [costs] * max(1, args.steps)
        # inert copies that pad UNet calls up to their bucket: work the rank does for nothing
        rp = [padded_work([lays[j].n_boxes for j in sh] * (max(1, args.group) if args.workload != "lmd_v0.1" else 1),
                          max(1, args.max_batch), max(1, min(args.max_batch_guided, args.max_batch)), LMDSampler.BUCKETS,
                          args.num_inference_steps) for sh in shares]
        my_pad_frac = sum(p for _, p in rp) / max(sum(r + p for r, p in rp), 1e-9)
        lane_cost = [0.0] * n_lanes

        def dry(lane, job):
            time.sleep(1e-4 * sum(job))
            lane_cost[lane.index] += sum(job)
        with LanePool([Lane(i, None) for i in range(n_lanes)]) as lp:
            lp.map(dry, jobs)
        dt = ldist.max_over_ranks(time.perf_counter() - t0)
        loads = ldist.gather_floats(float(my_cost))
        pads = ldist.gather_floats(float(my_pad_frac))
        csum = ldist.sum_over_ranks(float(ws.arena16.float().abs().sum()))
        ldist.shutdown()
        if rank == 0:
            print(json.dumps(dict(metric="dryrun", n_gpus=world, rccl_ranks=world, images=n_total,
                                  weight_broadcast_s=round(bcast_s, 4), per_rank_cost=loads, lanes_per_gpu=n_lanes,
                                  per_rank_padded_work_frac=[round(p, 4) for p in pads],
                                  images_per_unet_call=dict(unguided_max=args.max_batch, guided_max=args.max_batch_guided),
                                  rank0_lane_cost=[round(c, 1) for c in lane_cost],
                                  rank0_cpus=pinned, torch_threads=torch.get_num_threads(),
                                  weights_identical=abs(csum / world - float(ws.arena16.float().abs().sum())) < 1e-3,
                                  max_s=dt)))
        return

    pinned = ldist.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), lanes=max(1, args.lanes))
    from lgd_amd import ops
    from lgd_amd.pipeline import backward_guidance_generate_batch, lmd_generate_batch, lmd_plus_generate_batch
    from lgd_amd.sampler import LMDSampler
    from lgd_amd.scheduler import DDIMScheduler
    from lgd_amd.unet import UNetEngine
    from lgd_amd.vae import make_hip_vae
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.spawn:
        ldist.init(backend="nccl")
    # rank 0 materialises the weights; everyone else receives the two arenas over RCCL/xGMI
    mb = max(1, args.max_batch)
    mbg = max(1, min(args.max_batch_guided, mb))
    eng = UNetEngine(cfg, dev, weights.synth_state_dict(cfg, 0) if rank == 0 else None, max_text_batch=max(32, 2 * mb))
    bcast_s = ldist.broadcast_weights(eng.w, src=0, force=args.spawn) if (world > 1 or args.spawn) else 0.0
    from lgd_amd.lanes import LanePool, make_lanes
    side = 8 * cfg.sample_size
    if args.sam and (args.no_decode or args.workload == "backward_guidance"):
        raise SystemExit("--sam needs the decoded single-object images of LMD+ (no --no-decode, not backward_guidance)")

    def make_refiner():
        import transformers
        from lgd_amd.sam_refine import SamRefiner, wrap_sam
        torch.manual_seed(0)
        hf_sam = transformers.SamModel(transformers.SamConfig())          # architecture + parameter names; random weights
        with torch.no_grad():
            for name, prm in hf_sam.named_parameters():                   # HF zero-initialises the position tables
                if "pos" in name:
                    prm.normal_(0.0, 0.02)
        return SamRefiner(wrap_sam(hf_sam, "device", device=dev), height=side, width=side)

    def make_sampler(e):
        return LMDSampler(e, DDIMScheduler(prediction_type=cfg.prediction_type),
                          vae=None if args.no_decode else make_hip_vae(dev), max_batch=mb, max_batch_guided=mbg)

    # lanes: independent pipelines on their own HIP streams sharing one copy of the weights; with one lane this is
    # the plain sequential loop on a side stream
    lanes = make_lanes(eng, lanes_for(args.workload, len(lays), max(1, args.lanes), args.min_layouts_per_lane), make_sampler)
    for ln in lanes:
        ln.extras["refiner"] = make_refiner() if args.sam else None
    sm = lanes[0].sampler
    lane_pool = LanePool(lanes, device=dev)

    def one_step(lane, sub, n_steps=T, **kw):
        if not sub:
            return []
        if args.workload == "backward_guidance":       # generation/backward_guidance.py:46-49 defaults
            return backward_guidance_generate_batch(lane.sampler, sub, num_inference_steps=n_steps, height=side,
                                                    width=side, decode=not args.no_decode)
        if args.workload == "lmd":                     # generation/lmd.py:215-256 defaults (per-box guidance ON)
            kw = dict(dict(frozen_step_ratio=0.5, so_center_box=True, align_with_overall_bboxes=True), **kw)
            outs = lmd_generate_batch(lane.sampler, sub, num_inference_steps=n_steps, decode=not args.no_decode,
                                      mask_refiner=lane.extras["refiner"], **kw)
            for o in outs:                              # iterations of the per-box stage count as work of the image
                o["guidance_iters"] += sum(o["so_guidance_iters"])
                o["guidance_iters_fuser_on"] = 0
            return outs
        return lmd_plus_generate_batch(lane.sampler, sub, num_inference_steps=n_steps, decode=not args.no_decode,
                                       mask_refiner=lane.extras["refiner"], **kw)

    # jobs of one step: the whole batch (batch4 / backward_guidance: successive steps overlap across lanes) or, for
    # the prompt-set workload, one cost-balanced share of this rank's layouts per lane
    if args.workload == "lmd_v0.1" and len(lanes) > 1:
        shares = partition_by_cost([layout_cost(l.n_boxes, T) for l in lays], len(lanes))
        step_jobs = [[lays[j] for j in sh] for sh in shares]
    else:
        step_jobs = [lays]
    group = (args.group if args.group >= 0 else 1) if len(step_jobs) == 1 else 1

    def job_sizes(n_steps):
        """Steps per lane job.  group = g > 0: g consecutive steps per job (the last one takes the rest).  group = 0
        (auto): the steps are split EVENLY over the lanes in as few rounds as possible with at most --group-max steps per
        job (20 steps on 2 lanes, at most 5 per job: 4 jobs of 5; 4 steps: 2 jobs of 2), so that no lane idles at the end."""
        if n_steps <= 0:
            return []
        if group == 0:
            n_lanes = len(lanes)
            rounds = -(-n_steps // (n_lanes * max(1, args.group_max)))
            n_jobs = min(n_steps, rounds * n_lanes)
            return [n_steps // n_jobs + (1 if i < n_steps % n_jobs else 0) for i in range(n_jobs)]
        return [min(group, n_steps - i) for i in range(0, n_steps, group)]

    def jobs_of(n_steps):
        """The lane jobs of n_steps benchmark steps: the layouts of the steps a job takes at once share its UNet calls."""
        if len(step_jobs) > 1:
            return step_jobs * n_steps
        return [lays * g for g in job_sizes(n_steps)]

    # launch plans, GEMM kernel attributes and captured hipGraphs of every batch bucket this rank will use are built
    # BEFORE the timed barrier even with --warmup 0, lane by lane: a 2-step pass over the same layouts has the same
    # batch composition, and the sampler's device state (hence its graphs) does not depend on the step count
    t_pre = time.perf_counter()
    pre_kw = {} if args.workload == "backward_guidance" else dict(overall_max_index_step=2, frozen_step_ratio=0.5)
    if args.workload == "lmd":
        pre_kw["max_index_step"] = 2
    for k in range(len(lanes)):
        subs = sorted({len(j): j for j in jobs_of(args.steps) + jobs_of(args.warmup)}.values(), key=len) if len(step_jobs) == 1 else [step_jobs[k]]
        lane_pool.map(lambda lane, sub: one_step(lane, sub, 2, **pre_kw), subs, pin=[k] * len(subs))
    torch.cuda.synchronize()
    prebuild_s = time.perf_counter() - t_pre
    if args.warmup:
        lane_pool.map(one_step, jobs_of(args.warmup))
    for ln in lanes:
        ln.sampler.pass_counts.clear()
    ldist.barrier()
    torch.cuda.synchronize()
    clock = ClockSampler(local_rank)
    t0 = time.perf_counter()
    it_on = it_all = 0
    with clock:
        for outs in lane_pool.map(one_step, jobs_of(args.steps)):
            it_all += sum(o["guidance_iters"] for o in outs)
            it_on += sum(o["guidance_iters_fuser_on"] for o in outs)
        torch.cuda.synchronize()
    busy = time.perf_counter() - t0
    ldist.barrier()
    dt = ldist.max_over_ranks(time.perf_counter() - t0)
    per_rank_busy = ldist.gather_floats(busy)
    tot_on, tot_all = ldist.sum_over_ranks(float(it_on)), ldist.sum_over_ranks(float(it_all))
    tot_boxes = ldist.sum_over_ranks(float(my_boxes))
    n_images = args.steps * n_total
    lane_pool.close()
    pass_counts = {}
    for ln in lanes:
        for k_, v_ in ln.sampler.pass_counts.items():
            pass_counts[k_] = pass_counts.get(k_, 0) + v_
    ldist.shutdown()            # all ranks together, right after the last collective; the rest is rank-0 local
    if rank != 0:
        return
    iters_on, iters_off = tot_on / n_images, (tot_all - tot_on) / n_images       # per image, job-wide means
    mean_boxes = tot_boxes / n_total
    # ---- roofline of the dominant kernel.  The timed region replays captured hipGraphs (one launch per
    # UNet call), so individual kernels cannot be bracketed there; right after it, the very same plans are
    # launched eagerly on the same stream with HIP events around every GEMM/attention launch, and the
    # per-pass times are weighted by how often the timed region actually ran each pass (LMDSampler.pass_counts).
    roofline = None
    if not args.no_roofline and pass_counts:
        counts = pass_counts
        my_images = max(args.steps * len(lays), 1)
        agg, tag_agg, shape_agg = {}, {}, {}
        reps = 2
        mains = sorted({nb for (k, f, nb) in counts if k == "main"})
        guides = sorted({nb for (k, f, nb) in counts if k == "guide"})
        for kind, fz, nb_, fn in sm.profile_passes(cfg.sample_size, T, cfg.use_gated_attention, main_batches=mains,
                                                   guide_batches=guides,
                                                   ratio_energy=args.workload == "backward_guidance"):
            n_runs = counts.get((kind, fz, nb_), 0)
            if not n_runs:
                continue
            fn()
            torch.cuda.synchronize()
            prof = ops.LaunchProfiler(max_records=100000)
            ops.PROFILER = prof
            for _ in range(reps):
                fn()
            ops.PROFILER = None
            w = n_runs / reps / my_images                                 # per image of this rank
            for dst, summ in ((agg, prof.summary()), (tag_agg, prof.summary(by_tag=True)),
                              (shape_agg, prof.summary(by_shape=True))):
                for k, v in summ.items():
                    a = dst.setdefault(k, dict(ms=0.0, flops=0.0, bytes=0.0, n=0.0, raw_ms=0.0, raw_n=0))
                    a["ms"] += v["ms"] * w
                    a["flops"] += v["flops"] * w
                    a["bytes"] += v["bytes"] * w
                    a["n"] += v["n"] * w
                    a["raw_ms"] += v["ms"]
                    a["raw_n"] += v["n"]
        if args.shape_profile and shape_agg:
            os.makedirs(os.path.dirname(os.path.abspath(args.shape_profile)), exist_ok=True)
            with open(args.shape_profile, "w") as fh:
                json.dump({k: dict(ms_per_image=round(v["ms"], 3), launches_per_image=round(v["n"], 1),
                                   us_per_launch=round(v["raw_ms"] * 1e3 / max(v["raw_n"], 1), 2),
                                   tflops=round(v["flops"] / max(v["ms"] * 1e-3, 1e-12) / 1e12, 1))
                           for k, v in sorted(shape_agg.items(), key=lambda kv: -kv[1]["ms"])}, fh, indent=1)
        if agg:
            # dominant kernel = the MFMA-bound kernel family with the most time (the HBM-bound ones — norms, copies —
            # are listed in all_kernels with their GB/s; none of them comes near the GEMM / attention families)
            name, a = max(((k, v) for k, v in agg.items() if v["flops"] > 0), key=lambda kv: kv[1]["ms"])
            ach = a["flops"] / (a["ms"] * 1e-3)
            # HBM bytes per launch come from a committed PMC summary of THIS command's short form (PMC passes serialise
            # every dispatch: they cannot run inside the timed region) — the source file is named in the record
            traffic, traffic_note, traffic_source = None, "no PMC summary for this kernel under profiles/", None
            for fname in ("r06_bench_traffic_pmc.json", "r05_bench_traffic_pmc.json", "r04_bench_traffic_pmc.json",
                          "r03_bench_traffic_pmc.json", "r02_bench_traffic_pmc.json"):
                tpath = os.path.join(ROOT, "profiles", fname)
                if not os.path.exists(tpath):
                    continue
                tj = json.load(open(tpath))
                ent = tj.get("kernels", {}).get(name.split(" ")[0])
                if ent:
                    traffic, traffic_note, traffic_source = ent["hbm_bytes_per_launch"], tj.get("method", ""), f"profiles/{fname}"
                    break
            # the dominant kernel's heaviest SHAPE, with the HBM bytes one launch of exactly that shape moved (PMC passes on
            # that shape alone: tools/gpu_session.sh evidence -> profiles/r06_gemm_pmc_summary.json) against its algorithmic
            # bytes — the family average above mixes shapes and cannot be set against anything
            dominant_shape = None
            cand = [(k, v) for k, v in shape_agg.items() if k.split(" | ")[0] == name and " | " in k]
            if cand:
                import re
                sk, sv = max(cand, key=lambda kv: kv[1]["ms"])
                dominant_shape = dict(shape=sk.split(" | ")[1], ms_per_image=round(sv["ms"], 2), launches_per_image=round(sv["n"], 1),
                                      avg_launch_us=round(sv["raw_ms"] * 1e3 / max(sv["raw_n"], 1), 1),
                                      tflops=round(sv["flops"] / max(sv["ms"] * 1e-3, 1e-12) / 1e12, 1))
                m = re.match(r"M(\d+)_N(\d+)_K(\d+)_t(\d+)_.*_e(\d+)_", dominant_shape["shape"])
                ppath = os.path.join(ROOT, "profiles", "r06_gemm_pmc_summary.json")
                if m and os.path.exists(ppath):
                    key = "M%s_N%s_K%s_t%s" % m.groups()[:4] + ("_geglu" if int(m.group(5)) & 1 else "")
                    ent = json.load(open(ppath)).get("shapes", {}).get(key)
                    if ent and "traffic" in ent:
                        dominant_shape.update(traffic=ent["traffic"], algorithmic_bytes=ent["algorithmic_bytes"],
                                              traffic_ratio=ent["traffic_ratio"], mfma_busy_share=ent.get("mfma_busy_share"),
                                              traffic_source="profiles/r06_gemm_pmc_summary.json")
            gem = [v for k, v in agg.items() if k.startswith("gemm")]
            gemm_tf = sum(v["flops"] for v in gem) / max(sum(v["ms"] for v in gem) * 1e-3, 1e-12) / 1e12 if gem else None
            ap_ = tag_agg.get("attn_path")
            roofline = dict(bound="mfma", kernel=name, achieved=round(ach / 1e12, 2), peak=MFMA_PEAK_F16 / 1e12,
                            unit="TFLOP/s", frac=round(ach / MFMA_PEAK_F16, 4), traffic=traffic,
                            # weighted like est_ms_per_image (each pass by how often the timed region ran it), so that
                            # avg_launch_us x launches_per_image = est_ms_per_image
                            avg_launch_us=round(a["ms"] * 1e3 / a["n"], 2),
                            launches_per_image=round(a["n"], 1), est_ms_per_image=round(a["ms"], 1),
                            traffic_note=traffic_note, traffic_source=traffic_source, dominant_shape=dominant_shape,
                            rocprof_summary=ROCPROF_SUMMARY,
                            method="HIP events around each launch, eager replay of the benchmark's plans right after "
                                   "the timed region, ONE launch sequence alone on the GPU (the timed region itself "
                                   "replays hipGraphs, on config.lanes_per_gpu streams side by side — kernels of "
                                   "different lanes overlap there, so per-kernel durations can only be taken here); "
                                   "weights = the timed region's own pass counts",
                            # the second kernel by time: the two GEMM tile families trade places between runs and rounds
                            runner_up=(lambda k2, a2: dict(kernel=k2, achieved=round(a2["flops"] / (a2["ms"] * 1e-3) / 1e12, 2),
                                                          frac=round(a2["flops"] / (a2["ms"] * 1e-3) / MFMA_PEAK_F16, 4),
                                                          est_ms_per_image=round(a2["ms"], 1)))(
                                *sorted(((k, v) for k, v in agg.items() if v["flops"] > 0), key=lambda kv: -kv[1]["ms"])[1])
                            if sum(1 for v in agg.values() if v["flops"] > 0) > 1 else None,
                            all_gemm_tflops=round(gemm_tf, 1) if gemm_tf else None,
                            attention_path=(dict(what="q/k/v/out projections + SDPA (self, GLIGEN fuser, cross)",
                                                 ms_per_image=round(ap_["ms"], 1),
                                                 tflops=round(ap_["flops"] / (ap_["ms"] * 1e-3) / 1e12, 1),
                                                 frac_of_mfma_peak=round(ap_["flops"] / (ap_["ms"] * 1e-3) / MFMA_PEAK_F16, 4),
                                                 # the kernel that carries 60 % of the path's time, against what its
                                                 # instruction mix allows (the 40 % target reads against THIS, not 2.5 PF)
                                                 d40_self_attention=dict(
                                                     measured_tflops=(lambda v: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v else None)(
                                                         agg.get("attn_self_kernel d=40")),
                                                     **d40_attention_bounds()))
                                            if ap_ else None),
                            attention_backward=(lambda b_: dict(what="flash backward (dQ, dK/dV) + cross-attention dQ of the guidance "
                                                                     "iterations; 10 B H Sq Sk d per self-attention",
                                                                ms_per_image=round(b_["ms"], 1),
                                                                tflops=round(b_["flops"] / (b_["ms"] * 1e-3) / 1e12, 1))
                                                )(tag_agg["attn_path_bwd"]) if tag_agg.get("attn_path_bwd") else None,
                            # every entry point of a step is profiled (GEMMs incl. their split-K reduction, attention
                            # forward and backward, norms, energy, update kernels, plan-internal copies): the sum is the
                            # image's kernel time on one launch sequence
                            all_kernels_ms_per_image=round(sum(v["ms"] for v in agg.values()), 1),
                            all_kernels={k: (dict(ms_per_image=round(v["ms"], 1), launches_per_image=round(v["n"]),
                                                  tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)) if v["flops"] > 0 else
                                             dict(ms_per_image=round(v["ms"], 1), launches_per_image=round(v["n"]),
                                                  gb_per_s=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)))
                                         for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])})
    tf = None
    if args.config == "sd14_gligen" and args.workload != "backward_guidance":
        tf = algorithmic_tflop(mean_boxes, T, beta, iters_on, iters_off)
    elif args.config == "sd21" and args.workload == "backward_guidance":
        tf = T * TF_SD21_MAIN + (iters_on + iters_off) * TF_SD21_GUIDE
    elif args.config == "sd15" and args.workload == "lmd":
        # LMD = (N+1) x 50 plain CFG calls + every guidance iteration taken (per-box AND overall stage)
        tf = (mean_boxes + 1) * T * TF_MAIN_OFF + (iters_on + iters_off) * TF_GUIDE_OFF
    what = (f"{args.layouts} cached 2-box layouts/GPU/step" if args.workload != "lmd_v0.1" else
            f"{n_total} layouts of the lmd_v0.1 cache (0-5 boxes, mean {mean_boxes:.2f}) cost-balanced over {world} rank(s)")
    method = {"backward_guidance": "layout-guidance baseline (generation/backward_guidance.py), ",
              "lmd": "training-free LMD stage 2 (generation/lmd.py defaults: per-box AND overall cross-attention guidance, "
                     "max_index_step 30, reference-attention transfer, frozen_step_ratio 0.5, so_center_box + align_with_overall_bboxes), "}.get(args.workload, "LMD+ stage 2, ")
    arch = {"sd14_gligen": "SD1.4+GLIGEN architecture", "sd21": "SD2.1-768 architecture, v-prediction",
            "sd15": "SD1.5 architecture"}.get(args.config, args.config)
    metric = {"backward_guidance": f"images/sec (50-step SD2.1 {side}^2, backward guidance)",
              "lmd": "images/sec (50-step SD1.5 512^2, training-free LMD guidance)"}.get(
                  args.workload, "images/sec (50-step SD1.5 512^2, LMD+ guidance)")
    res = dict(metric=metric, value=round(n_images / dt, 4),
               unit="images/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(dt * 1e3 / args.steps, 1), higher_is_better=True,
               scaling="strong" if args.workload == "lmd_v0.1" else "weak",
               vs_baseline=None, dtype="fp16", data="synthetic",
               config=dict(workload=f"{method}{what} (lmd_v0.1 cache), {T} DDIM steps, {side}x{side}, {args.config} "
                                    f"({arch}, seeded random weights), VAE decodes "
                                    f"{'excluded' if args.no_decode else 'included'}"
                                    + ("; per-box masks refined by SAM (sam-vit-base architecture, random weights, "
                                       "HIP model + device-side processor)" if args.sam else "; per-box masks = box masks"),
                           mask_refinement="sam" if args.sam else "box",
                           layouts_per_gpu=(args.layouts if args.workload in ("batch4", "lmd") else round(n_total / world, 2)),
                           # how the step's generations are packed into UNet calls: a lane job = the layouts of
                           # `steps_per_lane_job` steps; their per-box / overall generations are chunked into calls of at
                           # most this many images (CFG batch = 2x), padded to a bucket of LMDSampler.BUCKETS
                           steps_per_lane_job=job_sizes(args.steps),
                           images_per_unet_call=dict(unguided_max=lanes[0].sampler.max_batch,
                                                     guided_max=lanes[0].sampler.max_batch_guided,
                                                     timed_region={f"{k_}{'+fuser' if f_ else ''} x{nb_}": n_
                                                                   for (k_, f_, nb_), n_ in sorted(pass_counts.items())}),
                           num_inference_steps=T, parallelism=f"dp{world}", rccl_ranks=world,
                           guidance_iters_per_image=round(iters_on + iters_off, 2),
                           guidance_iters_fuser_on=round(iters_on, 2),
                           algorithmic_tflop_per_image=round(tf, 3) if tf else None,
                           lanes_per_gpu=len(lanes), gemm_tuning=lanes[0].engine.tuning_mode or ops.current_tuning_mode(),
                           weight_broadcast_s=round(bcast_s, 3), prebuild_s=round(prebuild_s, 2),
                           per_rank_busy_s=[round(b, 3) for b in per_rank_busy],
                           per_rank_idle_s=[round(max(dt - b, 0.0), 3) for b in per_rank_busy],
                           sustained_gfx_clock=clock.summary(),
                           host=dict(torch_threads=1, rank0_cpu_affinity=(f"{pinned[0]}-{pinned[-1]}" if pinned else "unpinned"))),
               roofline=roofline)
    if tf:
        res["config"]["whole_path_frac_of_mfma_peak"] = round(tf * 1e12 * (n_images / dt) / world / MFMA_PEAK_F16, 4)
    if world == 1 and not args.no_cpu_baseline:
        try:
            gens = 1.0 if args.workload == "backward_guidance" else mean_boxes + 1
            res["cpu_baseline"] = cpu_baseline(cfg, gens, T, beta, iters_on, iters_off)
        except Exception as e:  # the baseline is reported, never gating
            res["cpu_baseline"] = dict(value=None, unit="images/s", cores=os.cpu_count(), kind="port",
                                       sample=f"failed: {e}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
