"""CPU suite: the N>1 path (one process per GPU in production) with world_size-2 gloo processes:
rank 0 packs the weights, the two arenas are broadcast, layouts are sharded round-robin with their global
index preserved (seeds derive from it), and the timing reduction is a max over ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lgd_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from lgd_amd import dist as ldist, weights
    from lgd_amd.weightstore import WeightStore
    from lgd_amd.pipeline import CachedLayout
    ldist.init(backend="gloo")
    cfg = weights.CONFIGS["tiny_gligen"]
    ws = WeightStore(cfg, "cpu")
    if rank == 0:
        ws.load_state_dict(weights.synth_state_dict(cfg, 0))
    secs = ldist.broadcast_weights(ws, src=0, chunk_bytes=8 << 20)
    layouts = [[("a cat", [10 * i, 20, 100, 120]), ("a dog", [300, 40 + i, 90, 100])] for i in range(7)]
    mine = ldist.shard(layouts)
    lays = [CachedLayout.synthetic(cfg, gb, index=i) for i, gb in mine]
    t = ldist.max_over_ranks(1.0 + rank)
    total = ldist.sum_over_ranks(float(len(mine)))
    torch.save(dict(sum16=float(ws.arena16.float().abs().sum()), sum32=float(ws.arena32.abs().sum()),
                    gate=ws.scalars.get("mid_block.attentions.0.transformer_blocks.0.fuser.alpha_attn"),
                    idx=[i for i, _ in mine], seeds=[(l.bg_seed, l.fg_seed_start) for l in lays],
                    noise0=lays[0].overall_cond[0, 0, :4].tolist(), t=t, total=total, secs=secs),
               os.path.join(out_dir, f"r{rank}.pt"))
    ldist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_layout_sharding_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world))
    # identical replicas after the broadcast (rank 1 never saw a state dict)
    assert r0["sum16"] == r1["sum16"] > 0 and r0["sum32"] == r1["sum32"] > 0
    assert abs(r0["gate"] - 0.7615941762924194) < 1e-6 and r0["gate"] == r1["gate"]
    # disjoint, complete, index-preserving partition; seeds follow the GLOBAL index (generate.py:226-229)
    assert r0["idx"] == [0, 2, 4, 6] and r1["idx"] == [1, 3, 5]
    assert r1["seeds"][0] == (1, 1 + 123456789)
    assert r0["t"] == r1["t"] == 2.0 and r0["total"] == r1["total"] == 7.0


def test_per_item_results_do_not_depend_on_partition():
    """A layout's synthetic text side and seeds depend only on its global index, so the 1-rank and
    2-rank runs produce the same per-prompt inputs (byte-identical)."""
    from lgd_amd import dist as ldist, weights
    from lgd_amd.pipeline import CachedLayout
    cfg = weights.CONFIGS["tiny"]
    layouts = [[("a cat", [10 * i, 20, 100, 120])] for i in range(5)]
    single = {i: CachedLayout.synthetic(cfg, gb, index=i) for i, gb in ldist.shard(layouts, 0, 1)}
    for r in range(2):
        for i, gb in ldist.shard(layouts, r, 2):
            lay = CachedLayout.synthetic(cfg, gb, index=i)
            assert torch.equal(lay.overall_cond, single[i].overall_cond) and lay.bg_seed == single[i].bg_seed


def test_bench_self_spawns_ranks_and_balances_the_prompt_set():
    """`python bench.py --gpus 2` with no rendezvous in the environment must start 2 ranks itself
    (torch.distributed.run), broadcast the weight arenas, partition the lmd_v0.1 prompt set by cost and
    report from rank 0 — exercised on CPU (gloo) through --cpu-dryrun."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-dryrun", "--workload",
                          "lmd_v0.1", "--prompts", "40"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == r["rccl_ranks"] == 2 and r["images"] == 40
    assert r["weight_broadcast_s"] > 0 and r["weights_identical"]
    a, b = r["per_rank_cost"]
    assert abs(a - b) / max(a, b) <= 0.05                     # cost = algorithmic TFLOP per layout, LPT-balanced
    # inside a rank the layouts are shared out over its lanes (host threads here, HIP streams on the GPU) the same way;
    # 20 layouts per rank = two lanes of ten (bench.lanes_for: a lane per 8 layouts, at most --lanes)
    lc = r["rank0_lane_cost"]
    assert r["lanes_per_gpu"] == len(lc) == 2 and abs(sum(lc) - a) < 1.0 and max(lc) / (sum(lc) / 2) <= 1.1
    assert max(r["per_rank_padded_work_frac"]) <= 0.10


def test_cost_partition_is_complete_balanced_and_deterministic():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cache = bench.load_cache()
    sel = bench.select_prompts(cache, 100)
    assert len(sel) == len(set(sel)) == 100
    n_boxes = [len(cache[i]["gen_boxes"]) for i in sel]
    assert min(n_boxes) == 0 and max(n_boxes) == 5             # all four prompt categories are present
    # cost model = algorithmic TFLOP of a layout: (N + 1) generations + the overall stage's guidance iterations, which
    # only layouts WITH boxes have (an empty layout is one unguided generation)
    costs = [bench.layout_cost(n) for n in n_boxes]
    assert abs(costs[n_boxes.index(0)] - (20 * bench.TF_MAIN_ON + 30 * bench.TF_MAIN_OFF)) < 1e-6
    assert abs(bench.layout_cost(2) - 359.8) < 0.2
    for world in (1, 2, 4, 8):
        parts = bench.partition_by_cost(costs, world)
        assert sorted(i for p in parts for i in p) == list(range(100))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)
        assert max(loads) / (sum(loads) / world) <= 1.05, (world, loads)     # >= 95 % of linear by load alone
        assert parts == bench.partition_by_cost(costs, world)
    # per-image work of the default 2-box LMD+ image with all 65 iterations: SURVEY.md 8(d) "<= 359.8 TF"
    assert abs(bench.algorithmic_tflop(2, 50, 0.4, 55, 10) - 359.8) < 0.2


def _dryrun8(prompts):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--cpu-dryrun", "--workload",
                          "lmd_v0.1", "--prompts", str(prompts), "--lanes", "4"], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_eight_rank_dryrun_balances_ranks_and_pins_hosts():
    """BASELINE config[3] shape without a node: 8 ranks over the 100-prompt lmd_v0.1 selection through
    `bench.py --gpus 8 --cpu-dryrun` (gloo): every rank's algorithmic load within 5 % of the mean (the >= 6x of the
    north star needs >= 0.75); lanes per rank DERIVED FROM THE WORK (12-13 layouts per rank: one lane, not four lanes of
    three layouts each); the work spent on inert copies that pad UNet calls to their bucket <= 10 % on every rank; one
    torch intra-op thread per rank and a CPU share per rank."""
    r = _dryrun8(100)
    assert r["n_gpus"] == 8 and r["images"] == 100 and r["weights_identical"]
    loads = r["per_rank_cost"]
    assert len(loads) == 8 and max(loads) / (sum(loads) / 8) <= 1.05, loads
    assert r["lanes_per_gpu"] == 1 and len(r["rank0_lane_cost"]) == 1
    pads = r["per_rank_padded_work_frac"]
    assert len(pads) == 8 and max(pads) <= 0.10, pads
    assert r["torch_threads"] == 1
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:
        assert r["rank0_cpus"] is not None and len(r["rank0_cpus"]) >= min(ncpu // 8, 5) and r["rank0_cpus"][0] == sorted(os.sched_getaffinity(0))[0]


def test_eight_rank_dryrun_whole_cache_uses_four_lanes_per_rank():
    """The strong-scaling set: all 400 layouts of the lmd_v0.1 cache (`--prompts 400`) over 8 ranks = 50 per
    rank, enough for four lanes each; ranks and lanes balanced, padded work <= 10 %."""
    r = _dryrun8(400)
    assert r["images"] == 400 and r["lanes_per_gpu"] == 4
    loads = r["per_rank_cost"]
    assert max(loads) / (sum(loads) / 8) <= 1.02, loads
    lc = r["rank0_lane_cost"]
    assert len(lc) == 4 and abs(sum(lc) - loads[0]) < 1.0 and max(lc) / (sum(lc) / 4) <= 1.05, lc
    assert max(r["per_rank_padded_work_frac"]) <= 0.10, r["per_rank_padded_work_frac"]


def test_plan_chunks_bounds_padding_and_covers_every_image():
    """sampler.plan_chunks: every image lands in exactly one call, calls are bucket-sized and within the cap, and no call
    carries more than max_pad inert copies (5 images are 4 + 1, not 8 with 3 copies)."""
    import lgd_amd  # noqa: F401
    from lgd_amd.sampler import LMDSampler, plan_chunks
    B = LMDSampler.BUCKETS
    for cap in (1, 3, 4, 8, 16, 32):
        for n in range(0, 70):
            ch = plan_chunks(n, cap, B)
            assert sum(c for c, _ in ch) == n
            for c, b in ch:
                assert 1 <= c <= b <= cap and b in B and (b - c) <= 0.25 * b, (n, cap, ch)
    assert plan_chunks(5, 8, B) == [(4, 4), (1, 1)] and plan_chunks(7, 8, B) == [(7, 8)]
    assert plan_chunks(20, 16, B) == [(16, 16), (4, 4)] and plan_chunks(0, 8, B) == []


def test_pin_rank_shares_are_disjoint_when_cores_suffice(monkeypatch):
    from lgd_amd import dist as ldist
    got = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: got.__setitem__("cpus", list(cpus)))
    n0 = torch.get_num_threads()
    try:
        shares = [ldist.pin_rank(r, 8, lanes=4) for r in range(8)]
        assert torch.get_num_threads() == 1
        assert all(len(s) == 8 for s in shares) and sorted(c for s in shares for c in s) == list(range(64))
        # fewer cores than lanes + 1 per rank: overlapping windows of lanes + 1 cores, never an empty set
        monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)))
        shares = [ldist.pin_rank(r, 8, lanes=4) for r in range(8)]
        assert all(len(s) == 5 and set(s) <= set(range(8)) for s in shares)
        assert ldist.pin_rank(0, 1, lanes=4) is None                 # one rank per node: the scheduler is left alone
    finally:
        torch.set_num_threads(n0)


def test_init_needs_a_port_from_the_launcher(monkeypatch):
    from lgd_amd import dist as ldist
    import pytest
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        ldist.init(backend="gloo")
