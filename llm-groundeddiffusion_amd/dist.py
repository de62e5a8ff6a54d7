"""Multi-GPU: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm).

The path shards embarrassingly: every cached layout is an independent run(), so
ranks take disjoint layout subsets and there is NO collective inside the denoising loop.  The only
communication is (a) one broadcast of the two packed weight arenas from rank 0 at start-up — a
pipelined ring/tree over xGMI, bound by the per-link bandwidth (~153 GB/s/link): 2.1 GB of
SD1.4+GLIGEN fp16 weights (+dgrad copies) land in tens of milliseconds — and (b) scalar
reductions for timing / counters.  The reference has no counterpart (it runs independent OS
processes over prompt windows: generate.py:23-25,243-250).
"""
import os
import time

import torch
import torch.distributed as dist


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def init(backend=None):
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        # No fixed default port (two jobs on one host would collide on it).  A launcher (torch.distributed.run, the
        # driver, bench.respawn) always names the port; only a single-process group can pick its own.
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise RuntimeError("MASTER_PORT is not set: ranks of one job must be given the same rendezvous port by "
                               "their launcher (python -m torch.distributed.run --master-port P ...)")
        os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend)


def pin_rank(local_rank: int, local_world: int, lanes: int = 1):
    """Host-side hygiene of one rank among `local_world` on a node: (1) torch intra-op threads = 1 — the host code of
    a rank is launch-bound Python plus tiny ATen CPU ops, and 8 ranks x 4 lane threads x an OpenMP pool of every core
    would thrash the host; (2) CPU affinity = this rank's contiguous share of the cores it was allowed to use (at
    least lanes + 1: one per lane thread + the main thread), so ranks do not migrate over each other's caches / NUMA
    nodes.  Returns the core list (None where the platform has no sched_setaffinity).  The CPU-baseline leg of
    bench.py (rank 0, N = 1 only) widens the thread count again for itself."""
    torch.set_num_threads(1)
    if not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // local_world
    if per < 1:
        return None                                           # fewer cores than ranks: leave the scheduler alone
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    if len(mine) < min(lanes + 1, len(cpus)):
        # not enough cores for lane threads + main thread in an exclusive share: overlap with the neighbours
        lo = max(0, min(local_rank * per, len(cpus) - (lanes + 1)))
        mine = cpus[lo:lo + lanes + 1]
    os.sched_setaffinity(0, mine)
    return mine


def shutdown():
    """Tear the process group down (local operation; no rank is inside a collective when it is called)."""
    if dist.is_initialized():
        dist.destroy_process_group()


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float) -> float:
    if not dist.is_initialized():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float) -> float:
    if not dist.is_initialized():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(x: float):
    """[x of rank 0, x of rank 1, ...] on every rank (per-rank busy time / load reports)."""
    if not dist.is_initialized():
        return [float(x)]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def broadcast_weights(store, src=0, chunk_bytes=256 << 20, force=False) -> float:
    """Replicates a WeightStore: two flat arenas, broadcast in large chunks (fewer, larger collectives suit the
    point-to-point xGMI links), ALL chunks issued as non-blocking collectives and waited for once (round 6: they used to
    be blocking calls one after another on the default stream).  force: issue the collectives with one rank as well (the
    1-GPU rehearsal of the N-GPU path, bench.py --spawn).  Returns seconds."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return 0.0
    if store.arena16.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    works = []
    for arena in (store.arena16, store.arena32):
        n = arena.numel()
        step = max(1, chunk_bytes // arena.element_size())
        for off in range(0, n, step):
            works.append(dist.broadcast(arena[off:off + step], src=src, async_op=True))
    for w in works:
        w.wait()
    if store.arena16.is_cuda:
        torch.cuda.synchronize()
    store.refresh_scalars()
    return time.perf_counter() - t0


def shard(items, r=None, w=None):
    """Round-robin partition preserving the global index (seeds derive from it: generate.py:226-229),
    so per-item outputs do not depend on the number of ranks."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return [(i, it) for i, it in enumerate(items) if i % w == r]
