"""Two denoising pipelines on two HIP streams of one GPU (lgd_amd/lanes.py) against the same jobs run one after
another on a single pipeline: the engines share one copy of the weights, everything a run writes is per lane, so
the results must be BIT-identical — any shared scratch (split-K workspace, arena, time / text tables, the process-wide
CPU generator the reference seeds per draw) would show up here as a mismatch."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import lgd_amd  # noqa: E402,F401
from lgd_amd import weights  # noqa: E402
from lgd_amd.lanes import LanePool, make_lanes  # noqa: E402
from lgd_amd.pipeline import CachedLayout, lmd_plus_generate_batch  # noqa: E402
from lgd_amd.sampler import LMDSampler  # noqa: E402
from lgd_amd.scheduler import DDIMScheduler  # noqa: E402
from lgd_amd.unet import UNetEngine  # noqa: E402
from lgd_amd.vae import make_hip_vae  # noqa: E402

L = 32
BOXES = [("a white deer", [74, 177, 183, 235]), ("a gray bear", [314, 193, 189, 216]), ("a red ball", [40, 60, 200, 180]),
         ("a blue cup", [250, 40, 120, 110])]


def _jobs(cfg):
    # different box counts -> different batch buckets, plans and guidance-loop lengths per job
    return [[CachedLayout.synthetic(cfg, BOXES[:2], 3), CachedLayout.synthetic(cfg, BOXES[2:3], 5)],
            [CachedLayout.synthetic(cfg, BOXES[1:4], 7)],
            [CachedLayout.synthetic(cfg, [], 9), CachedLayout.synthetic(cfg, BOXES[:1], 11)],
            [CachedLayout.synthetic(cfg, BOXES[:2], 13), CachedLayout.synthetic(cfg, BOXES[2:4], 15)]]


def test_two_lanes_equal_one_lane_bit_for_bit(dev):
    cfg = weights.CONFIGS["tiny_gligen"]
    eng = UNetEngine(cfg, dev, weights.synth_state_dict(cfg, 0))
    make = lambda e: LMDSampler(e, DDIMScheduler(), vae=make_hip_vae(dev))
    kw = dict(num_inference_steps=6, height=8 * L, width=8 * L, decode=True, overall_loss_threshold=0.0,
              overall_max_index_step=3, overall_max_iter=[2])
    run = lambda lane, lays: lmd_plus_generate_batch(lane.sampler, lays, **kw)
    jobs = _jobs(cfg)

    lanes = make_lanes(eng, 2, make)
    assert lanes[1].engine is not eng and lanes[1].engine.w is eng.w          # one copy of the parameters
    assert lanes[0].stream.cuda_stream != lanes[1].stream.cuda_stream
    with LanePool(lanes, device=dev) as pool:
        first = pool.map(run, jobs)
        again = pool.map(run, list(reversed(jobs)))[::-1]                      # other job -> lane assignment
    with LanePool(lanes[:1], device=dev) as pool:
        serial = pool.map(run, jobs)

    n = 0
    for a, b, c in zip(first, again, serial):
        assert len(a) == len(b) == len(c)
        for ra, rb, rc in zip(a, b, c):
            assert ra["guidance_iters"] == rb["guidance_iters"] == rc["guidance_iters"]
            for k in ("latents", "image"):
                ta, tb, tc = (torch.as_tensor(r[k]) for r in (ra, rb, rc))
                assert torch.isfinite(ta.float()).all()
                assert torch.equal(ta, tc), f"two lanes vs one lane differ in {k}"
                assert torch.equal(tb, tc), f"lane assignment changed {k}"
            n += 1
    assert n == 7


def test_job_error_reaches_the_caller_and_the_pool_survives(dev):
    cfg = weights.CONFIGS["tiny_gligen"]
    eng = UNetEngine(cfg, dev, weights.synth_state_dict(cfg, 0))
    lanes = make_lanes(eng, 2, lambda e: LMDSampler(e, DDIMScheduler()))
    kw = dict(num_inference_steps=2, height=8 * L, width=8 * L, decode=False, overall_max_index_step=1)
    with LanePool(lanes, device=dev) as pool:
        def bad(lane, x):
            if x == 1:
                raise RuntimeError("boom")
            return lmd_plus_generate_batch(lane.sampler, [CachedLayout.synthetic(cfg, BOXES[:1], 3)], **kw)
        with pytest.raises(RuntimeError, match="boom"):
            pool.map(bad, [0, 1, 0])
        out = pool.map(bad, [0, 0])
        assert torch.equal(out[0][0]["latents"], out[1][0]["latents"])


def test_bench_self_spawn_path_with_rccl_on_one_gpu():
    """the first real N-GPU run must not also be the first run of bench.py's self-spawn path.  `--gpus 1
    --spawn` goes through `respawn` (python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1,
    a free port), rank pinning, RCCL init with one rank, the non-blocking chunked arena broadcast, the barriers and the
    gather of per-rank times, on the full-width network — everything an 8-GPU run does except crossing xGMI."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "1", "--warmup", "0",
           "--num-inference-steps", "2", "--layouts", "1", "--lanes", "1", "--no-cpu-baseline", "--no-roofline", "--no-decode"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["value"] > 0
    assert line["config"]["weight_broadcast_s"] > 0.0                    # the collectives were really issued
    assert len(line["config"]["per_rank_busy_s"]) == 1
