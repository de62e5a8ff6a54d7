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


def test_dpm_solver_first_order_is_ddim_and_second_order_is_exact_on_a_linear_data_prediction():
    """[ext] DPMSolverMultistepScheduler (models/models.py:46-47), restated: (a) with solver_order 1 every step equals the
    DDIM step between the same two timesteps (DPM-Solver++(1) IS DDIM); (b) the 2M rows reproduce the closed form of the
    update when the data prediction is linear in lambda — the case the second-order correction integrates exactly to
    its order: x0(lambda) = u + lambda w  =>  exact solution  x(l') = (s'/s) x - a' (e^{-h} - 1) u - a' ((e^{-h} - 1) l' + h) w... checked
    through the defining relation D1 = w * (l_t - l_prev) / r = w h; (c) lower_order_final and the last-step target."""
    import torch
    from lgd_amd.scheduler import DDIMScheduler, DPMSolverMultistepScheduler
    d1 = DPMSolverMultistepScheduler(solver_order=1)
    d1.set_timesteps(10)
    dd = DDIMScheduler(steps_offset=0)
    assert d1.timesteps.tolist() == [999, 899, 799, 699, 599, 500, 400, 300, 200, 100]
    x, e = torch.randn(1, 4, 8, 8, generator=torch.manual_seed(0)), torch.randn(1, 4, 8, 8, generator=torch.manual_seed(1))
    ts = d1.timesteps.tolist()
    for i, t in enumerate(ts):
        tn = ts[i + 1] if i + 1 < len(ts) else 0
        a_t, a_p = float(dd.alphas_cumprod[t]), float(dd.alphas_cumprod[tn])
        x0 = (x - (1 - a_t) ** 0.5 * e) / a_t ** 0.5
        ref = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * e
        out, x0h = d1.step_host(e, i, x)
        assert float((out - ref).abs().max()) < 2e-6 and float((x0h - x0).abs().max()) < 1e-5
    d2 = DPMSolverMultistepScheduler()
    d2.set_timesteps(50)
    rows = d2.multistep_rows()
    assert rows[0][4] == 0.0 and all(r[4] != 0.0 for r in rows[1:])          # 50 >= 15 steps: the last one stays second order
    ts = d2.timesteps.tolist()
    for i in range(1, 50):
        a_t, s_t, l_t = d2._als(ts[i])
        a_n, s_n, l_n = d2._als(ts[i + 1] if i + 1 < 50 else 0)
        _, _, l_p = d2._als(ts[i - 1])
        h = l_n - l_t
        c0, c1, A, B, C = rows[i]
        # x0 = u + lambda w sampled at l_t and l_p:  D0 = u + l_t w,  D1 = (D0 - D0_prev) / r = w h
        u, w = 0.7, -0.3
        x0_t, x0_p = u + l_t * w, u + l_p * w
        import math
        got = B * x0_t + C * x0_p
        want = -a_n * math.expm1(-h) * (x0_t + 0.5 * w * h)
        assert abs(got - want) < 1e-9 * max(1.0, abs(want))
        assert abs(A - s_n / s_t) < 1e-12
    d3 = DPMSolverMultistepScheduler()
    d3.set_timesteps(10)
    assert d3.multistep_rows()[-1][4] == 0.0                                # fewer than 15 steps: first-order final step
