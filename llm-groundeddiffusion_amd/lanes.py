"""Several denoising pipelines on ONE GPU, each on its own HIP stream.

Why: half of the UNet's GEMM time is spent at the 16x16 / 8x8 levels, where a launch has fewer output tiles than the
chip has CUs (M = 256 ... 4096 rows: 0.3 - 1.9 waves of workgroups, 250 - 700 TF/s against 1000 - 1200 at the 64x64
level) and every kernel boundary drains the machine.  Those holes cannot be filled from inside one dependent launch
sequence — the next layer needs the previous one — but layouts are independent of each other (generate.py runs one
`run(spec)` after another), so a SECOND launch sequence working on other layouts can: two lanes on two streams finish
two steps of the default benchmark in 1.65 - 1.7x the time of one (measured with two processes before this module
existed: 1.329 -> 0.807 + 0.772 images/s on one MI355X).

A lane = one host thread + one HIP stream + one UNetEngine / LMDSampler / VAE with their own activation arena, time /
text / GLIGEN tables, split-K scratch, launch plans and captured hipGraphs.  The parameters are shared
(`UNetEngine(weights=other.w)`: read-only after load).  Nothing inside a lane knows about the others; results are
bit-identical to running the same jobs one after another on a single lane (`tests/test_lanes_gpu.py`).

Process-wide state the lanes do share, and how it is kept safe:
  * torch's default CPU generator (the reference seeds it per draw, latents.py:7-18) — `hostprep.RNG_LOCK`;
  * hipGraph capture — while a stream of the process is capturing, ROCm 7.2 rejects plain work of OTHER host threads
    (a pageable host-to-device copy failed with "operation not permitted when stream is capturing" even under
    capture_error_mode="thread_local"), so a capture is an exclusive section: `GATE.exclusive()` waits until every
    other lane is parked at a `GATE.checkpoint()` (the samplers call it once per denoising step and guidance
    iteration) or between jobs, and holds them there until the graph exists (`sampler.HipGraph`);
  * split-K scratch of calls that bring none — thread-local (`ops.workspace`);
  * the interpreter lock: a lane that comes back from a device wait must not sit out CPython's default 5 ms switch
    interval behind the other lane's host code (65 guidance syncs per image) — the pool lowers it to 0.2 ms.
"""
import contextlib
import queue
import sys
import threading
from typing import Any, Callable, List, Optional, Sequence

import torch


class _Gate:
    """Exclusive sections among the host threads that are inside a LanePool job."""

    def __init__(self):
        self.cv = threading.Condition()
        self.active = set()        # idents of threads inside a job
        self.parked = set()        # of those: waiting at a checkpoint or for their own exclusive section
        self.owner = None
        self.waiters = 0

    def enter_job(self):
        with self.cv:
            while self.owner is not None or self.waiters:
                self.cv.wait()
            self.active.add(threading.get_ident())

    def exit_job(self):
        with self.cv:
            self.active.discard(threading.get_ident())
            self.parked.discard(threading.get_ident())
            self.cv.notify_all()

    def checkpoint(self):
        """Safe point of a lane: no HIP call of this thread is in flight.  Free unless a capture is pending."""
        if self.owner is None and not self.waiters:
            return
        me = threading.get_ident()
        with self.cv:
            if me not in self.active:
                return
            self.parked.add(me)
            self.cv.notify_all()
            while self.owner is not None or self.waiters:
                self.cv.wait()
            self.parked.discard(me)

    @contextlib.contextmanager
    def exclusive(self):
        me = threading.get_ident()
        with self.cv:
            if self.owner == me:               # nested
                nested = True
            else:
                nested = False
                self.waiters += 1
                if me in self.active:
                    self.parked.add(me)
                self.cv.notify_all()
                while self.owner is not None or any(t not in self.parked for t in self.active if t != me):
                    self.cv.wait()
                self.owner = me
                self.waiters -= 1
                self.parked.discard(me)
        try:
            yield
        finally:
            if not nested:
                with self.cv:
                    self.owner = None
                    self.cv.notify_all()


GATE = _Gate()


class Lane:
    """One pipeline.  `sampler` (and through it the engine and the VAE) must only ever be driven from this lane's
    thread while the pool is running."""

    def __init__(self, index: int, sampler, stream=None, extras: Optional[dict] = None):
        self.index = index
        self.sampler = sampler
        self.stream = stream
        self.extras = extras or {}

    @property
    def engine(self):
        return self.sampler.eng


class LanePool:
    """Runs jobs `fn(lane, item)` on N lanes; a lane takes the next job as soon as it finished its last (the jobs of
    a denoising workload differ in length: box counts, guidance exits).  `map` returns the results in item order
    and re-raises the first exception of a job (the plugin boundary's RuntimeError convention, generate.py:391-396)."""

    def __init__(self, lanes: Sequence[Lane], device=None):
        if not lanes:
            raise ValueError("a LanePool needs at least one lane")
        self.lanes = list(lanes)
        self.device = device
        self._inbox: List["queue.Queue"] = [queue.Queue() for _ in self.lanes]    # one per lane: its next job
        self._threads: List[threading.Thread] = []
        self._closed = False
        self._map_lock = threading.Lock()                 # one map() at a time: it owns every lane while it runs
        self._switch_interval = None
        if len(self.lanes) > 1 and sys.getswitchinterval() > 2e-4:
            self._switch_interval = sys.getswitchinterval()      # restored by close()
            sys.setswitchinterval(2e-4)
        for k, lane in enumerate(self.lanes):
            t = threading.Thread(target=self._worker, args=(k, lane), name=f"lgd-lane-{lane.index}", daemon=True)
            t.start()
            self._threads.append(t)

    def __len__(self):
        return len(self.lanes)

    # ------------------------------------------------------------------------------------------------------------
    def _worker(self, k: int, lane: Lane):
        if self.device is not None:
            torch.cuda.set_device(self.device)            # the current device is per host thread
        from . import ops
        mode = getattr(getattr(lane.sampler, "eng", None), "tuning_mode", None)
        with ops.tuning(mode):                            # launches the lane describes outside plans (VAE, text)
            self._serve(k, lane)

    def _serve(self, k: int, lane: Lane):
        while True:
            task = self._inbox[k].get()
            if task is None:
                return
            fn, item, idx, done = task
            GATE.enter_job()
            try:
                if lane.stream is not None:
                    with torch.cuda.stream(lane.stream):
                        res = (True, fn(lane, item))
                        lane.stream.synchronize()         # the job's device work is finished when map() returns
                else:
                    res = (True, fn(lane, item))
            except BaseException as e:                    # noqa: BLE001 - delivered to the caller of map()
                res = (False, e)
            finally:
                GATE.exit_job()
            done.put((k, idx, res))

    def map(self, fn: Callable[[Lane, Any], Any], items: Sequence[Any], pin: Optional[Sequence[Optional[int]]] = None) -> List[Any]:
        """pin[i] = position (in this pool) of the lane job i must run on; None / omitted = whichever lane is free
        first, in item order."""
        if self._closed:
            raise RuntimeError("LanePool is closed")
        items = list(items)
        n = len(self.lanes)
        if pin is not None:
            pin = list(pin)
            if len(pin) != len(items) or any(p is not None and not (0 <= p < n) for p in pin):
                raise ValueError(f"pin must name a lane 0..{n - 1} (or None) for each of the {len(items)} jobs")
        box: List[Any] = [None] * len(items)
        with self._map_lock:
            done: "queue.Queue" = queue.Queue()
            free_jobs = [i for i in range(len(items)) if pin is None or pin[i] is None]
            own_jobs = [[i for i in range(len(items)) if pin is not None and pin[i] == k] for k in range(n)]
            busy = [False] * n
            left = len(items)

            def feed(k):
                if busy[k]:
                    return
                idx = own_jobs[k].pop(0) if own_jobs[k] else (free_jobs.pop(0) if free_jobs else None)
                if idx is not None:
                    busy[k] = True
                    self._inbox[k].put((fn, items[idx], idx, done))

            for k in range(n):
                feed(k)
            while left:
                k, idx, res = done.get()
                box[idx] = res
                busy[k] = False
                left -= 1
                feed(k)
        out = []
        for ok, val in box:
            if not ok:
                raise val
            out.append(val)
        return out

    def each(self, fn: Callable[[Lane], Any]) -> List[Any]:
        """fn(lane) once on EVERY lane (pre-building plans / graphs, warm-up), concurrently."""
        return self.map(lambda lane, _: fn(lane), [None] * len(self.lanes), pin=list(range(len(self.lanes))))

    def close(self):
        if self._closed:
            return
        self._closed = True
        for q in self._inbox:
            q.put(None)
        for t in self._threads:
            t.join(timeout=30)
        if self._switch_interval is not None:
            sys.setswitchinterval(self._switch_interval)
            self._switch_interval = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def make_lanes(engine, n_lanes: int, make_sampler: Callable[[Any], Any]) -> List[Lane]:
    """Lane 0 drives `engine` itself; lanes 1.. get engines that share its parameters.  make_sampler(engine) builds
    the lane's LMDSampler (scheduler, VAE, batch limits: whatever the caller wants, one fresh set per lane)."""
    from .unet import UNetEngine
    lanes = []
    for i in range(max(1, int(n_lanes))):
        eng = engine if i == 0 else UNetEngine(engine.cfg, engine.device, text_len=engine.text_len,
                                               max_text_batch=engine.max_text_batch, weights=engine.w)
        if int(n_lanes) > 1:
            # GEMM tiles / split-K chosen for a shared GPU: a property of the lane's ENGINE (its plans), not of the
            # process — an engine built later for single-sequence use still gets the "latency" table
            eng.tuning_mode = "throughput"
        stream = torch.cuda.Stream(device=engine.device) if engine.device.type == "cuda" else None
        lanes.append(Lane(i, make_sampler(eng), stream))
    return lanes
