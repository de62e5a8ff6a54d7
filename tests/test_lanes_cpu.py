"""Host logic of lgd_amd/lanes.py (no GPU): result order, dynamic hand-out, pinning, error delivery, and the lock
around the reference's seed-then-draw idiom on the process-wide CPU generator."""
import threading
import time

import pytest
import torch

import lgd_amd  # noqa: F401
from lgd_amd import hostprep
from lgd_amd.lanes import Lane, LanePool


def test_results_in_item_order_and_free_lane_takes_next_job():
    with LanePool([Lane(i, None) for i in range(2)]) as pool:
        seen = []

        def job(lane, x):
            time.sleep(0.05 if x == 0 else 0.005)
            seen.append((lane.index, x))
            return x * x
        assert pool.map(job, [0, 1, 2, 3, 4]) == [0, 1, 4, 9, 16]
        # the lane that got the long job 0 must not have been handed every other job as well
        assert len({ln for ln, _ in seen}) == 2
        long_lane = next(ln for ln, x in seen if x == 0)
        assert sum(1 for ln, _ in seen if ln != long_lane) >= 3


def test_pin_each_and_errors():
    with LanePool([Lane(i, None) for i in range(3)]) as pool:
        assert pool.each(lambda lane: lane.index) == [0, 1, 2]
        assert pool.map(lambda lane, x: (lane.index, x), ["a", "b", "c"], pin=[2, 2, 0]) == [(2, "a"), (2, "b"), (0, "c")]
        with pytest.raises(ValueError):
            pool.map(lambda lane, x: x, [1], pin=[3])
        with pytest.raises(ZeroDivisionError):
            pool.map(lambda lane, x: 1 // x, [1, 0, 1])
        assert pool.map(lambda lane, x: x + 1, [1, 2]) == [2, 3]          # still alive
    with pytest.raises(RuntimeError):
        pool.map(lambda lane, x: x, [1])                                    # closed


def test_seeded_noise_is_atomic_across_threads():
    """latents.py:7-18 re-seeds the PROCESS-WIDE generator before each draw; two lanes doing that at once must each
    still get the noise of their own seed."""
    want = {s: hostprep.seeded_noise(s, 4, 32, 32).clone() for s in range(8)}
    bad = []

    def worker(seeds):
        for _ in range(30):
            for s in seeds:
                if not torch.equal(hostprep.seeded_noise(s, 4, 32, 32), want[s]):
                    bad.append(s)
    ts = [threading.Thread(target=worker, args=(list(range(k, 8, 2)),)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


def test_exclusive_section_waits_for_the_other_lanes_checkpoints():
    """hipGraph capture must not overlap ANY HIP call of another host thread on ROCm 7.2 (lanes.py): an exclusive
    section starts only when every other lane inside a job is parked at a checkpoint, and holds them there."""
    from lgd_amd.lanes import GATE
    log = []

    def job(lane, kind):
        if kind == "capture":
            time.sleep(0.01)
            with GATE.exclusive():
                with GATE.exclusive():                     # nested (a capture that builds another graph)
                    t0 = time.perf_counter()
                    time.sleep(0.03)
                    log.append(("x", t0, time.perf_counter()))
            return 1
        for _ in range(15):
            GATE.checkpoint()
            t0 = time.perf_counter()
            time.sleep(0.004)                              # "HIP calls" between two checkpoints
            log.append(("w", t0, time.perf_counter()))
        return 0
    with LanePool([Lane(i, None) for i in range(3)]) as pool:
        assert pool.map(job, ["work", "capture", "work", "capture", "work"]) == [0, 1, 0, 1, 0]
    xs = [e for e in log if e[0] == "x"]
    ws = [e for e in log if e[0] == "w"]
    assert len(xs) == 2 and len(ws) == 45
    for _, a, b in xs:
        assert not [w for w in ws if w[1] < b and w[2] > a], "work of another lane overlapped an exclusive section"
    assert xs[0][2] <= xs[1][1] or xs[1][2] <= xs[0][1]
