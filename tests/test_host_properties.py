"""Size-independent properties of the host-side rules (hypothesis): they hold for any box / offset / schedule,
not just for the golden cases."""
import math

import torch
from hypothesis import given, settings, strategies as st

import lgd_amd  # noqa: F401
from lgd_amd import hostprep as hp
from lgd_amd.scheduler import DDIMScheduler

unit = st.floats(min_value=0.0, max_value=1.0, allow_nan=False)


@settings(max_examples=200, deadline=None)
@given(x0=unit, y0=unit, w=st.floats(0.01, 0.6), h=st.floats(0.01, 0.6), k=st.integers(-3, 3), n=st.sampled_from([8, 16, 32, 64]))
def test_pixel_extent_is_translation_invariant(x0, y0, w, h, k, n):
    """The reason for rounding start and extent separately (utils/utils.py:62): moving a box by whole pixels
    never changes its pixel size (as long as it stays inside the grid)."""
    x0, y0 = x0 * (1 - w), y0 * (1 - h)
    a = hp.scale_proportion([x0, y0, x0 + w, y0 + h], n, n)
    assert 0 <= a[0] <= a[2] <= n and 0 <= a[1] <= a[3] <= n
    xs = x0 + k / n
    if 0 <= xs and xs + w <= 1 and a[2] < n and a[0] > 0:
        b = hp.scale_proportion([xs, y0, xs + w, y0 + h], n, n)
        if 0 < b[0] and b[2] < n:
            assert abs((b[2] - b[0]) - (a[2] - a[0])) <= 1          # +-1 only through float noise in (hi-lo)*n
    m = hp.proportion_to_mask([x0, y0, x0 + w, y0 + h], n, n)
    assert int(m.sum()) == (a[2] - a[0]) * (a[3] - a[1])


@settings(max_examples=100, deadline=None)
@given(dx=st.integers(-20, 20), dy=st.integers(-20, 20), seed=st.integers(0, 100))
def test_shift_is_a_translation_with_zero_fill(dx, dy, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((2, 3, 16, 24), generator=g)
    y = hp.shift_tensor(x, dx, dy)
    H, W = 16, 24
    for (r, c) in [(0, 0), (5, 7), (15, 23), (8, 12)]:
        rs, cs = r - dy, c - dx                                       # source pixel of output pixel (r, c)
        inside = 0 <= rs < H and 0 <= cs < W
        assert torch.equal(y[..., r, c], x[..., rs, cs] if inside else torch.zeros(2, 3))
    back = hp.shift_tensor(y, -dx, -dy)                               # round trip restores the part never pushed out
    keep = torch.zeros(H, W, dtype=torch.bool)
    keep[max(-dy, 0):max(H - max(dy, 0), 0), max(-dx, 0):max(W - max(dx, 0), 0)] = True
    assert torch.equal(back[..., keep], x[..., keep]) and float(back[..., ~keep].abs().sum()) == 0.0
    att = torch.randn((2, 16, 24, 5), generator=g)                    # attention-map form [..., h, w, tokens]
    assert torch.equal(hp.shift_tensor(att, dx, dy, ignore_last_dim=True).movedim(-1, 0),
                       hp.shift_tensor(att.movedim(-1, 0), dx, dy))


@settings(max_examples=100, deadline=None)
@given(fx=st.floats(-0.9, 0.9), fy=st.floats(-0.9, 0.9))
def test_normalised_shift_moves_every_level_by_the_same_fraction(fx, fy):
    """8x8-grid quantisation: a 64x64 latent and a 16x16 / 8x8 map shifted by the same normalised offset stay
    registered (the 64-grid shift is exactly 8x / 4x the coarse ones)."""
    probe = {}
    for n in (8, 16, 64):
        x = torch.zeros(n, n)
        x[n // 2, n // 2] = 1.0
        y = hp.shift_tensor(x, fx, fy, offset_normalized=True)
        nz = y.nonzero()
        probe[n] = None if len(nz) == 0 else ((int(nz[0, 0]) - n // 2) * (64 // n), (int(nz[0, 1]) - n // 2) * (64 // n))
    if probe[8] is not None and probe[16] is not None and probe[64] is not None:
        assert probe[8] == probe[16] == probe[64]


@settings(max_examples=100, deadline=None)
@given(T=st.integers(2, 100), fa=st.integers(0, 110), rate=st.integers(1, 4))
def test_fast_schedule_and_dynamic_steps(T, fa, rate):
    sch = DDIMScheduler()
    sch.set_timesteps(T)
    full = [int(t) for t in sch.timesteps]
    ts = [int(t) for t in sch.fast_schedule(sch.timesteps, fa, rate)]
    assert ts[:min(fa, T)] == full[:min(fa, T)] and set(ts) <= set(full) and ts == sorted(ts, reverse=True)
    if fa >= T - 1:
        assert ts == full
    sizes = sch.dynamic_step_sizes(ts)
    for i, (t, s) in enumerate(zip(ts, sizes)):
        nxt = ts[i + 1] if i + 1 < len(ts) else -1
        gap = t - nxt
        assert s == 1000 // (1000 // gap) and s >= gap                # never undershoots the next timestep ...
        if 1000 % gap == 0 or gap * gap <= 1000:
            assert s == gap                                           # ... and lands on it for the usual divisors
    tab = sch.coef_table(7.5, "cpu", timesteps=ts, step_ratios=sizes)
    assert tab.shape == (len(ts), 4) and bool((tab[:, 0] > 0).all()) and bool((tab[:, 1] >= tab[:, 0]).all())
