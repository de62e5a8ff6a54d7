"""Host-side DDIM tables incl. the optional fast schedule against goldens
produced by the reference's own utils/schedule.py (oracle/make_golden_schedule.py)."""
import json
import os

import pytest
import torch

import lgd_amd  # noqa: F401
from lgd_amd.scheduler import DDIMScheduler

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedule_fast.json")


@pytest.mark.parametrize("case", json.load(open(GOLD)), ids=lambda c: f"T{c['T']}_fast{c['fast_after_steps']}")
def test_fast_schedule_tables_match_reference(case):
    sch = DDIMScheduler()
    sch.set_timesteps(case["T"])
    ts = sch.timesteps
    if case["fast_after_steps"] is not None:
        ts = sch.fast_schedule(ts, case["fast_after_steps"], case["fast_rate"])
    gold = case["steps"]
    assert [int(t) for t in ts] == [r[0] for r in gold]                      # integer work: exact
    sizes = sch.dynamic_step_sizes(ts)
    assert [int(t) - s for t, s in zip(ts, sizes)] == [r[1] for r in gold]
    tab = sch.coef_table(7.5, "cpu", timesteps=ts, step_ratios=sizes)
    ref = torch.tensor([[r[2], r[3]] for r in gold], dtype=torch.float32)
    assert torch.equal(tab[:, :2], ref)                                       # same fp32 table entries
    assert float(tab[0, 2]) == 7.5


def test_dynamic_sizes_equal_static_rule_on_the_plain_schedule():
    """Without a fast tail the per-step re-derivation changes nothing (only the last step differs in
    prev_t, and both land below 0 -> final_alpha_cumprod)."""
    for T in (50, 30, 25, 10, 7):
        sch = DDIMScheduler()
        sch.set_timesteps(T)
        a = sch.coef_table(7.5, "cpu")
        b = sch.coef_table(7.5, "cpu", step_ratios=sch.dynamic_step_sizes(sch.timesteps))
        assert torch.equal(a, b)
