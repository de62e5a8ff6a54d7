"""Per-kernel parity: each C-ABI entry point vs the plain PyTorch fp32 op sequence it replaces
(inputs rounded to fp16 first, reference computed in fp32).  Tolerances are stated per test."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import lgd_amd  # noqa: E402
from conftest import gate  # noqa: E402
from lgd_amd import ops  # noqa: E402

H16, F32 = torch.float16, torch.float32


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def rnd(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def pack_conv_w(w):  # [Cout,Cin,3,3] -> [Cout, 9*Cin] (ky,kx,ci)
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_geglu(w, b):  # [2n,K] -> 16-row value/gate interleave
    n = w.shape[0] // 2
    idx = []
    for j in range(n // 16):
        idx += list(range(16 * j, 16 * j + 16)) + list(range(n + 16 * j, n + 16 * j + 16))
    idx = torch.tensor(idx, device=w.device)
    return w[idx].contiguous(), (b[idx].contiguous() if b is not None else None)


@pytest.mark.parametrize("M,N,K,tile,splits", [
    (8192, 320, 320, 0, 1), (4100, 640, 640, 1, 1), (512, 1280, 1280, 2, 1), (128, 1280, 5120, 4, 4),
    (77, 320, 768, 5, 1), (154, 1280, 768, 0, 1), (30, 640, 768, 5, 3), (2048, 640, 2560, 3, 2),
    (64, 64, 32, 4, 1), (100, 100, 72, 2, 1),
    # LDS-DMA main loop (tile code 16 + t); K % 64 != 0 falls back to the register-staged loop
    (8192, 320, 320, 17, 1), (4100, 640, 640, 18, 1), (512, 1280, 1280, 19, 1), (128, 1280, 5120, 20, 4),
    (77, 320, 768, 21, 1), (30, 640, 768, 21, 3), (2048, 640, 2560, 19, 2), (100, 100, 72, 18, 1),
    (1000, 200, 64, 17, 1),
    # 8-wave 256-row tiles (25: 256x320, 26: 256x128)
    (8192, 320, 320, 25, 1), (4100, 640, 640, 25, 1), (2048, 640, 2560, 26, 2), (300, 1280, 1280, 26, 1),
    (515, 960, 192, 25, 1),
    # round 4, two-stage rings: 44 = 256x256 (eight waves of 64x128), 45 = 128x128 with two workgroups per CU
    (8192, 512, 320, 44, 1), (4100, 2560, 640, 44, 1), (300, 1280, 1280, 44, 1), (515, 1280, 1280, 45, 1), (4100, 640, 640, 45, 1),
    (8192, 320, 64, 44, 1), (8192, 320, 64, 45, 1),                      # one K tile: the ring's prologue alone
    # round 6, phase-split tiles: 46 = 256x256, 47 = 256x320 (ragged M / N: zero-filled rows through the buffer descriptor's
    # range check; one, two and three K tiles: prologue-only, one loop trip with both look-ahead tiles dead, odd trip counts;
    # split-K through fp32 partials)
    (8192, 512, 320, 46, 1), (4100, 2560, 640, 46, 1), (300, 1280, 1280, 46, 1), (8192, 320, 64, 46, 1), (515, 1000, 128, 46, 1),
    (8192, 320, 320, 47, 1), (4100, 640, 640, 47, 1), (300, 1280, 1280, 47, 1), (8192, 320, 64, 47, 1), (515, 968, 192, 47, 1),
    (2048, 640, 2560, 46, 2), (2048, 640, 2560, 47, 3), (300, 1280, 5120, 47, 4),
])
def test_gemm_plain(dev, M, N, K, tile, splits):
    x = rnd(M, K, dev=dev, seed=1).half()
    w = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).half()
    b = rnd(N, dev=dev, seed=3)
    r = rnd(M, N, dev=dev, seed=4).half()
    y = ops.linear(x, w, b, res=r, tile=tile, splits=splits, alpha=0.5)
    ref = (x.float() @ w.float().t() + b) * 0.5 + r.float()
    assert relerr(y, ref) < 3e-3


def test_gemm_out_f32_bias2(dev):
    M, N, K = 300, 320, 640
    x = rnd(M, K, dev=dev, seed=1).half()
    w = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).half()
    b, b2 = rnd(N, dev=dev, seed=3), rnd(N, dev=dev, seed=5)
    y = ops.linear(x, w, b, bias2=b2, out_f32=True)
    ref = x.float() @ w.float().t() + b + b2
    assert y.dtype == F32 and relerr(y, ref) < 1e-3


@pytest.mark.parametrize("M,dim,splits,tile", [(4096, 320, 1, 0), (300, 640, 1, 0), (128, 1280, 2, 0),
                                               (4096, 320, 1, 17), (300, 640, 1, 19), (128, 1280, 2, 20),
                                               (4100, 320, 1, 44), (300, 640, 1, 44), (4100, 320, 1, 45), (300, 640, 2, 45),
                                               (4100, 320, 1, 46), (300, 640, 1, 46), (300, 640, 2, 46)])
def test_gemm_geglu(dev, M, dim, splits, tile):
    inner = 4 * dim
    x = rnd(M, dim, dev=dev, seed=1).half()
    w = rnd(2 * inner, dim, dev=dev, seed=2, scale=dim ** -0.5).half()
    b = rnd(2 * inner, dev=dev, seed=3)
    wp, bp = pack_geglu(w, b)
    y = ops.linear(x, wp, bp, geglu=True, splits=splits, tile=tile)
    h = x.float() @ w.float().t() + b
    v, g = h.chunk(2, dim=-1)
    ref = v * F.gelu(g)
    assert y.shape == (M, inner) and relerr(y, ref) < 3e-3


@pytest.mark.parametrize("M,C,N,geglu,tile,splits", [
    (8192, 320, 960, False, 0, 1), (8192, 320, 2560, True, 0, 1), (2048, 640, 640, False, 0, 1),
    (300, 640, 5120, True, 0, 1), (128, 1280, 3840, False, 0, 2), (128, 1280, 10240, True, 0, 2),
    (4100, 320, 960, False, 33, 1), (4100, 640, 1920, False, 34, 1), (4100, 320, 2560, True, 34, 1),   # LDS epilogue, ragged M
    (515, 1280, 1280, False, 37, 1), (515, 320, 320, False, 2, 1), (515, 640, 640, False, 18, 1),
    (128, 1280, 1280, False, 20, 4),                                                                    # split-K, both reducers
    (4100, 320, 2560, True, 44, 1), (4100, 640, 1920, False, 44, 1), (515, 640, 1920, False, 45, 1), (515, 320, 2560, True, 45, 1),
    (4100, 320, 2560, True, 46, 1), (4100, 640, 1920, False, 46, 1), (4100, 320, 960, False, 47, 1), (515, 1280, 1280, False, 47, 2),
])
def test_gemm_rownorm_is_layernorm_then_linear(dev, M, C, N, geglu, tile, splits):
    """LGD_EPI_ROWNORM (ABI v8): statistics pass + GEMM on the raw rows with gamma-folded weights == LayerNorm
    followed by the linear layer (attention.py:185,206,223), on rows with a mean far from zero."""
    x = (rnd(M, C, dev=dev, seed=1) * 1.5 + rnd(M, 1, dev=dev, seed=7) * 2.0).half()
    w = rnd(N, C, dev=dev, seed=2, scale=C ** -0.5)
    b = rnd(N, dev=dev, seed=3)
    gm, bt = 1.0 + 0.3 * rnd(C, dev=dev, seed=4), 0.2 * rnd(C, dev=dev, seed=5)
    h = F.layer_norm(x.float(), (C,), gm, bt, 1e-5) @ w.half().float().t() + b
    if geglu:
        v, g = h.chunk(2, dim=-1)
        ref = v * F.gelu(g)
        wk, bk = pack_geglu(w, b)
    else:
        ref, wk, bk = h, w, b
    wln = (wk.half().float() * gm[None, :]).half()
    cs = wln.float().sum(1)
    bln = bk + wk.half().float() @ bt
    for in_launch in (True, False):
        stats = ops.layernorm_stats(x, C)
        mu, var = x.float().mean(1), x.float().var(1, unbiased=False)
        assert relerr(stats[:, 0], mu) < 1e-4 and relerr(stats[:, 1], (var + 1e-5).rsqrt()) < 1e-4
        old = ops.SPLITK_IN_LAUNCH
        ops.SPLITK_IN_LAUNCH = in_launch
        try:
            y = ops.linear(x, wln, bln, geglu=geglu, tile=tile, splits=splits, rowstat=stats, colsum=cs)
        finally:
            ops.SPLITK_IN_LAUNCH = old
        assert relerr(y, ref) < 4e-3, (in_launch, relerr(y, ref))
        if splits == 1:
            break


@pytest.mark.parametrize("dma", [0, 16])
@pytest.mark.parametrize("B,H,C0,C1,Cout,stride,ups,splits", [
    (2, 16, 64, 0, 64, 1, False, 1), (2, 32, 320, 0, 320, 1, False, 1),
    (1, 16, 128, 64, 128, 1, False, 1), (2, 8, 1280, 1280, 1280, 1, False, 8),
    (2, 16, 320, 0, 320, 2, False, 1), (1, 8, 640, 0, 640, 1, True, 1),
    (1, 10, 64, 0, 128, 1, False, 1), (1, 16, 64, 0, 64, 1, 2, 1),
    (2, 16, 8, 0, 32, 1, False, 1), (4, 64, 8, 0, 320, 1, False, 1),      # conv_in as an implicit GEMM: 8 input channels, K = 72
])
def test_conv3x3(dev, B, H, C0, C1, Cout, stride, ups, splits, dma):
    C = C0 + C1
    x = rnd(B, C, H, H, dev=dev, seed=1).half()
    w = rnd(Cout, C, 3, 3, dev=dev, seed=2, scale=(9 * C) ** -0.5).half()
    b = rnd(Cout, dev=dev, seed=3)
    xl = x.permute(0, 2, 3, 1).reshape(B * H * H, C).contiguous()
    x0 = xl[:, :C0].contiguous()
    x1 = xl[:, C0:].contiguous() if C1 else None
    xin = x.float()
    if ups == 1 or ups is True:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    elif ups == 2:
        z = torch.zeros(B, C, 2 * H, 2 * H, device=dev)
        z[:, :, ::2, ::2] = xin
        xin = z
    ref = F.conv2d(xin, w.float(), b, stride=stride, padding=1)
    Ho = ref.shape[-1]
    res = rnd(B * Ho * Ho, Cout, dev=dev, seed=7).half()
    y = ops.conv3x3(x0, pack_conv_w(w), B, H, H, x1=x1, bias=b, res=res, stride=stride,
                    ups=int(ups), splits=splits, tile=(dma + 3 if dma else 0))
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout) + res.float()
    assert relerr(y, ref) < 3e-3


@pytest.mark.parametrize("tile", [25, 26])
def test_conv3x3_8wave_tiles(dev, tile):
    B, H, C0, C1, Cout = 2, 32, 320, 320, 320
    C = C0 + C1
    x = rnd(B, C, H, H, dev=dev, seed=1).half()
    w = rnd(Cout, C, 3, 3, dev=dev, seed=2, scale=(9 * C) ** -0.5).half()
    b = rnd(Cout, dev=dev, seed=3)
    xl = x.permute(0, 2, 3, 1).reshape(B * H * H, C).contiguous()
    res = rnd(B * H * H, Cout, dev=dev, seed=7).half()
    y = ops.conv3x3(xl[:, :C0].contiguous(), pack_conv_w(w), B, H, H, x1=xl[:, C0:].contiguous(), bias=b, res=res,
                    tile=tile, splits=1)
    ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout) + res.float()
    assert relerr(y, ref) < 3e-3


@pytest.mark.parametrize("tile,B,H,C,Cout,splits", [(47, 5, 32, 320, 320, 1), (46, 3, 48, 128, 256, 1), (47, 2, 16, 1280, 640, 2),
                                                     (47, 9, 8, 640, 1280, 3), (46, 1, 64, 64, 320, 1)])
def test_conv3x3_phase_tiles(dev, tile, B, H, C, Cout, splits):
    """Round 6: 3x3 stride-1 convolution on the phase-split tiles — chunk-major K walk, padded taps and rows past M
    zero-filled by the buffer descriptor's range check (image borders, several images per tile at 8x8 / 16x16, a tile
    that straddles images, ragged M), fp32 partials for split-K."""
    x = rnd(B, C, H, H, dev=dev, seed=1).half()
    w = rnd(Cout, C, 3, 3, dev=dev, seed=2, scale=(9 * C) ** -0.5).half()
    b = rnd(Cout, dev=dev, seed=3)
    xl = x.permute(0, 2, 3, 1).reshape(B * H * H, C).contiguous()
    res = rnd(B * H * H, Cout, dev=dev, seed=7).half()
    y = ops.conv3x3(xl, pack_conv_w(w), B, H, H, bias=b, res=res, tile=tile, splits=splits)
    ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout) + res.float()
    assert relerr(y, ref) < 3e-3
    y2 = ops.conv3x3(xl, pack_conv_w(w), B, H, H, bias=b, res=res, tile=tile, splits=splits)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("M,N,K,tile", [
    (66017, 1000, 320, 34), (66017, 1000, 640, 33), (40000, 968, 192, 37), (70001, 320, 320, 38), (33000, 2560, 320, 35),
    (131072, 320, 64, 39), (20000, 1920, 576, 40),
])
def test_gemm_persistent_tile_loop_ragged(dev, M, N, K, tile):
    """More output tiles than resident workgroups and K <= 640: the 8-wave tiles run their persistent loop (one
    workgroup walks tiles b, b + grid, ... with the DMA ring crossing tile boundaries).  Ragged M and N: clamped rows
    in the last tiles of a walk, tile counts that are not a multiple of the grid, N tiles past the last column."""
    x = rnd(M, K, dev=dev, seed=1).half()
    w = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).half()
    b = rnd(N, dev=dev, seed=3)
    r = rnd(M, N, dev=dev, seed=4).half()
    y = ops.linear(x, w, b, res=r, tile=tile, splits=1)
    ref = x.float() @ w.float().t() + b + r.float()
    assert relerr(y, ref) < 3e-3
    y2 = ops.linear(x, w, b, res=r, tile=tile, splits=1)           # second launch: nothing stale between launches
    assert torch.equal(y, y2)


@pytest.mark.parametrize("B,H,Cout,tile", [(10, 48, 320, 37), (9, 64, 160, 38), (3, 96, 320, 34)])
def test_conv3x3_persistent_tile_loop(dev, B, H, Cout, tile):
    """3x3 convolution over 64 input channels (K = 576 <= 640) with more tiles than resident workgroups: the
    chunk-major K walk restarts on every output tile of the persistent loop."""
    C = 64
    x = rnd(B, C, H, H, dev=dev, seed=1).half()
    w = rnd(Cout, C, 3, 3, dev=dev, seed=2, scale=(9 * C) ** -0.5).half()
    b = rnd(Cout, dev=dev, seed=3)
    xl = x.permute(0, 2, 3, 1).reshape(B * H * H, C).contiguous()
    res = rnd(B * H * H, Cout, dev=dev, seed=7).half()
    y = ops.conv3x3(xl, pack_conv_w(w), B, H, H, bias=b, res=res, tile=tile, splits=1)
    ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout) + res.float()
    assert relerr(y, ref) < 3e-3


def test_conv_dgrad_identity(dev):
    """dgrad of a stride-1 conv = conv with flipped, transposed weights (how the engine calls it)."""
    B, H, Cin, Cout = 1, 16, 64, 128
    x = rnd(B, Cin, H, H, dev=dev, seed=1).requires_grad_(True)
    w = rnd(Cout, Cin, 3, 3, dev=dev, seed=2, scale=0.05).half()
    gy = rnd(B, Cout, H, H, dev=dev, seed=3).half()
    F.conv2d(x, w.float(), padding=1).backward(gy.float())
    wd = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()  # [Cin, Cout, 3, 3]
    gyl = gy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous()
    gx = ops.conv3x3(gyl, pack_conv_w(wd), B, H, H)
    ref = x.grad.permute(0, 2, 3, 1).reshape(-1, Cin)
    assert relerr(gx, ref) < 3e-3


def test_nchw_to_nhwc8(dev):
    x = rnd(3, 4, 16, 16, dev=dev, seed=1)
    y = ops.nchw_to_nhwc8(x)
    xl = x.permute(0, 2, 3, 1).reshape(-1, 4)
    assert torch.equal(y[:, :4], xl.half())
    assert torch.equal(y[:, 4:], (xl - xl.half().float()).half())          # rounding remainder in the spare channels
    assert float((y[:, :4].float() + y[:, 4:].float() - xl).abs().max()) < 1e-6


def test_conv_in_out(dev):
    B, L, C = 2, 64, 320
    x = rnd(B, 4, L, L, dev=dev, seed=1)
    w = rnd(C, 4, 3, 3, dev=dev, seed=2, scale=0.15).half()
    b = rnd(C, dev=dev, seed=3)
    y = ops.conv_in(x, pack_conv_w(w), b)
    ref = F.conv2d(x, w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, C)
    assert relerr(y, ref) < 2e-3
    h = rnd(B * L * L, C, dev=dev, seed=4).half()
    w2 = rnd(4, C, 3, 3, dev=dev, seed=5, scale=0.02).half()
    b2 = rnd(4, dev=dev, seed=6)
    o = ops.conv_out(h, pack_conv_w(w2), b2, B, L, out_scale=0.5)
    ref2 = F.conv2d(h.float().reshape(B, L, L, C).permute(0, 3, 1, 2), w2.float(), b2, padding=1) * 0.5
    assert relerr(o, ref2) < 1e-3


@pytest.mark.parametrize("B,HW,C0,C1,silu,eps", [
    (2, 4096, 320, 0, True, 1e-5), (2, 1024, 640, 320, True, 1e-5), (1, 256, 1280, 1280, True, 1e-5),
    (2, 64, 1280, 0, False, 1e-6), (1, 4096, 64, 0, True, 1e-5), (2, 256, 128, 64, False, 1e-6),
])
def test_groupnorm_fwd_bwd(dev, B, HW, C0, C1, silu, eps):
    C, G = C0 + C1, 32
    x = (rnd(B, HW, C, dev=dev, seed=1) * 2 + 0.5).half()
    gamma, beta = 1 + 0.2 * rnd(C, dev=dev, seed=2), 0.2 * rnd(C, dev=dev, seed=3)
    x0 = x[..., :C0].reshape(B * HW, C0).contiguous()
    x1 = x[..., C0:].reshape(B * HW, C1).contiguous() if C1 else None
    stats = torch.empty(B, G, 2, device=dev)
    y = ops.groupnorm(x0, B, HW, G, eps, gamma, beta, silu, x1=x1, stats=stats)
    xr = x.float().permute(0, 2, 1).clone().requires_grad_(True)  # (B, C, HW)
    ref = F.group_norm(xr, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    refl = ref.permute(0, 2, 1).reshape(B * HW, C)
    assert relerr(y, refl) < 3e-3
    gy = rnd(B * HW, C, dev=dev, seed=4).half()
    ref.backward(gy.float().reshape(B, HW, C).permute(0, 2, 1))
    gx0, gx1 = ops.groupnorm_bwd(gy, x0, B, HW, G, gamma, beta, silu, stats, x1=x1)
    gref = xr.grad.permute(0, 2, 1)
    assert relerr(gx0, gref[..., :C0].reshape(B * HW, C0)) < 5e-3
    if C1:
        assert relerr(gx1, gref[..., C0:].reshape(B * HW, C1)) < 5e-3


@pytest.mark.parametrize("B,HW,C0,C1,silu", [
    (2, 256, 1280, 0, True), (2, 256, 1280, 1280, True), (3, 256, 1280, 640, True),    # cpg 40, 80, 60 (two groups per workgroup)
    (2, 256, 640, 0, True), (2, 64, 1280, 1280, True), (5, 64, 1280, 0, False),          # cpg 20; the 8x8 level
    (2, 256, 320, 0, False), (1, 100, 960, 0, True), (2, 256, 128, 64, False),           # cpg 10 (4 groups), 30, 6 (concat inside a group)
    (2, 1024, 640, 320, True), (1, 1024, 1280, 640, True),                               # 32x32 with "gn_fused" raised to 1024
])
def test_groupnorm_one_launch_vs_torch_and_two_launch(dev, B, HW, C0, C1, silu):
    """The register-resident one-launch GroupNorm of the small maps (csrc/norm.hip gn_fused_kernel) against
    torch.group_norm and against the two-launch kernels, statistics included (the backward reads them)."""
    C, G, eps = C0 + C1, 32, 1e-5
    x = (rnd(B, HW, C, dev=dev, seed=1) * 2 + 3.0 * rnd(B, 1, C, dev=dev, seed=9)).half()      # per-channel offsets: mean >> std in places
    gamma, beta = 1 + 0.2 * rnd(C, dev=dev, seed=2), 0.2 * rnd(C, dev=dev, seed=3)
    x0 = x[..., :C0].reshape(B * HW, C0).contiguous()
    x1 = x[..., C0:].reshape(B * HW, C1).contiguous() if C1 else None
    st1, st2 = torch.zeros(B, G, 2, device=dev), torch.zeros(B, G, 2, device=dev)
    try:
        ops.set_option("gn_fused", max(HW, 256))
        y1 = ops.groupnorm(x0, B, HW, G, eps, gamma, beta, silu, x1=x1, stats=st1)
        ops.set_option("gn_fused", 0)
        y2 = ops.groupnorm(x0, B, HW, G, eps, gamma, beta, silu, x1=x1, stats=st2)
    finally:
        ops.set_option("gn_fused", 256)
    ref = F.group_norm(x.float().permute(0, 2, 1), G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    refl = ref.permute(0, 2, 1).reshape(B * HW, C)
    xg = x.float().reshape(B, HW, G, C // G)
    mean, var = xg.mean(dim=(1, 3)), xg.var(dim=(1, 3), unbiased=False)
    assert relerr(st1[..., 0], mean) < 1e-5 and relerr(st1[..., 1], (var + eps).rsqrt()) < 1e-5
    assert relerr(st2[..., 0], mean) < 1e-4 and relerr(st2[..., 1], (var + eps).rsqrt()) < 1e-3
    assert relerr(y1, refl) < 2e-3 and relerr(y2, refl) < 3e-3
    assert relerr(y1, y2) < 3e-3


@pytest.mark.parametrize("B,HW,C0,C1,silu", [
    (2, 256, 1280, 0, True), (4, 64, 1280, 1280, True), (2, 256, 1280, 1280, False),    # cpg 40, 80: 256- and 512-thread slabs
    (3, 64, 1280, 0, False), (2, 256, 640, 0, True), (2, 64, 1280, 640, True),          # cpg 20 (two groups per slab), 60
    (2, 100, 960, 0, True), (1, 256, 320, 0, True), (2, 256, 1280, 640, True),          # ragged map; cpg 10; a slab over 96 KB (two launches)
    (2, 4096, 320, 0, True), (2, 1024, 640, 320, True),                                  # large maps stay on the two-launch kernels
])
def test_groupnorm_bwd_slab_kernel_vs_torch_and_two_launch(dev, B, HW, C0, C1, silu):
    """Round 6: GroupNorm backward in ONE launch (csrc/norm.hip gn_bwd_slab_kernel: x and gy of the (image, groups) slab
    in the registers of one workgroup, taken for slabs of <= 96 KB) against torch and against the two-launch kernels it
    replaces (option "gn_slab" = 0), the accumulating form included; bit-reproducible."""
    C, G, eps = C0 + C1, 32, 1e-5
    x = (rnd(B, HW, C, dev=dev, seed=1) * 2 + 3.0 * rnd(B, 1, C, dev=dev, seed=9)).half()      # per-channel offsets: mean >> std in places
    gamma, beta = 1 + 0.2 * rnd(C, dev=dev, seed=2), 0.2 * rnd(C, dev=dev, seed=3)
    x0 = x[..., :C0].reshape(B * HW, C0).contiguous()
    x1 = x[..., C0:].reshape(B * HW, C1).contiguous() if C1 else None
    gy = rnd(B * HW, C, dev=dev, seed=4).half()
    acc0 = rnd(B * HW, C0, dev=dev, seed=5).half()
    acc1 = rnd(B * HW, C1, dev=dev, seed=6).half() if C1 else None
    res = []
    try:
        for slab in (1, 0):
            ops.set_option("gn_slab", slab)
            st = torch.zeros(B, G, 2, device=dev)
            y = ops.groupnorm(x0, B, HW, G, eps, gamma, beta, silu, x1=x1, stats=st)
            gx0, gx1 = ops.groupnorm_bwd(gy, x0, B, HW, G, gamma, beta, silu, st, x1=x1)
            a0, a1 = ops.groupnorm_bwd(gy, x0, B, HW, G, gamma, beta, silu, st, x1=x1, gx0=acc0.clone(),
                                       gx1=acc1.clone() if C1 else None, accumulate=True)
            res.append((y, st, gx0, gx1, a0, a1))
    finally:
        ops.set_option("gn_slab", 1)
    xr = x.float().permute(0, 2, 1).clone().requires_grad_(True)
    ref = F.group_norm(xr, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    ref.backward(gy.float().reshape(B, HW, C).permute(0, 2, 1))
    refl = ref.detach().permute(0, 2, 1).reshape(B * HW, C)
    gref = xr.grad.permute(0, 2, 1).reshape(B * HW, C)
    xg = x.float().reshape(B, HW, G, C // G)
    mean, var = xg.mean(dim=(1, 3)), xg.var(dim=(1, 3), unbiased=False)
    for tag, (y, st, gx0, gx1, a0, a1) in zip(("slab", "two-launch"), res):
        assert relerr(st[..., 0], mean) < 1e-4 and relerr(st[..., 1], (var + eps).rsqrt()) < 1e-3, tag
        assert relerr(y, refl) < 3e-3, tag
        assert relerr(gx0, gref[:, :C0]) < 5e-3, tag
        assert relerr(a0, gref[:, :C0] + acc0.float()) < 5e-3, tag
        if C1:
            assert relerr(gx1, gref[:, C0:]) < 5e-3 and relerr(a1, gref[:, C0:] + acc1.float()) < 5e-3, tag
    assert relerr(res[0][0], res[1][0]) < 3e-3 and relerr(res[0][2], res[1][2]) < 3e-3
    # bit-reproducible: fixed summation order, no atomics
    ops.set_option("gn_slab", 1)
    st = torch.zeros(B, G, 2, device=dev)
    y2 = ops.groupnorm(x0, B, HW, G, eps, gamma, beta, silu, x1=x1, stats=st)
    g2, _ = ops.groupnorm_bwd(gy, x0, B, HW, G, gamma, beta, silu, st, x1=x1)
    assert torch.equal(y2, res[0][0]) and torch.equal(st, res[0][1]) and torch.equal(g2, res[0][2])


@pytest.mark.parametrize("rows,C,ldx", [
    (65536, 320, 320), (16384, 640, 640), (8200, 1280, 1280), (32771, 320, 320), (13001, 640, 640),    # 8 / 16 / 32 lanes per row, ragged
    (30000, 320, 960), (130, 2560, 2560), (7, 2560, 2560),                                               # row stride; 64 lanes per row
    (4096, 1280, 1280), (515, 640, 640), (300, 1536, 1536), (77, 64, 64),                                # small maps stay on the row kernels
])
def test_layernorm_statistics_streaming_kernel(dev, rows, C, ldx):
    """Round 6: the statistics-only LayerNorm as a stream (csrc/norm.hip ln_stats_kernel: 8 .. 64 lanes share a row, DPP
    sums) against fp32 torch and against the one-wave-per-row kernels it replaces (option "ln_stream" = 0); rows past
    the end are not written."""
    x = (rnd(rows, ldx, dev=dev, seed=1) * 1.5 + rnd(rows, 1, dev=dev, seed=7) * 2.0).half()
    xv = x[:, :C]
    mu, var = xv.float().mean(1), xv.float().var(1, unbiased=False)
    got = []
    try:
        for stream in (1, 0):
            if not stream and C > 192 * 8:
                continue
            ops.set_option("ln_stream", stream)
            st = torch.full((rows + 8, 2), -7.0, device=dev)
            ops.layernorm_stats(x, C, stats=st, rows=rows, ldx=ldx)
            assert relerr(st[:rows, 0], mu) < 1e-5 and relerr(st[:rows, 1], (var + 1e-5).rsqrt()) < 1e-5, stream
            assert bool((st[rows:] == -7.0).all())
            got.append(st[:rows].clone())
    finally:
        ops.set_option("ln_stream", 1)
    if len(got) == 2:
        assert relerr(got[0], got[1]) < 2e-6


@pytest.mark.parametrize("rows,C", [(8192, 320), (2048, 640), (513, 1280), (30, 64)])
def test_layernorm_fwd_bwd(dev, rows, C):
    x = (rnd(rows, C, dev=dev, seed=1) * 1.5 + 0.3).half()
    gamma, beta = 1 + 0.2 * rnd(C, dev=dev, seed=2), 0.2 * rnd(C, dev=dev, seed=3)
    stats = torch.empty(rows, 2, device=dev)
    y = ops.layernorm(x, gamma, beta, stats=stats)
    xr = x.float().clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    assert relerr(y, ref) < 2e-3
    gy = rnd(rows, C, dev=dev, seed=4).half()
    ref.backward(gy.float())
    gx = ops.layernorm_bwd(gy, x, gamma, stats)
    assert relerr(gx, xr.grad) < 5e-3


def _attn_ref(q, k, v, scale):
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    p = s.softmax(-1)
    return torch.einsum("bhqk,bhkd->bhqd", p, v), p


@pytest.mark.parametrize("B,H,Sq,Sk,d", [
    (2, 8, 4096, 4096, 40), (2, 8, 1024, 1054, 80), (1, 8, 256, 286, 160), (2, 8, 64, 64, 160),
    (1, 5, 576, 576, 64), (2, 8, 100, 77, 8), (1, 8, 256, 256, 16), (1, 8, 64, 94, 32),
    (9, 8, 256, 286, 160), (5, 8, 300, 256, 160),        # round 6: > 256 (image, head, 64-query) blocks -> one 256-query workgroup per (image, head)
])
def test_attn_fwd_bwd(dev, B, H, Sq, Sk, d):
    C = H * d
    scale = d ** -0.5
    q = rnd(B, Sq, C, dev=dev, seed=1).half()
    k = rnd(B, Sk, C, dev=dev, seed=2).half()
    v = rnd(B, Sk, C, dev=dev, seed=3).half()
    o = torch.empty(B, Sq, C, device=dev, dtype=H16)
    lse = torch.empty(B, H, Sq, device=dev)
    ops.attn_fwd(q, k, v, o, B, H, Sq, Sk, d, scale, lse=lse)
    sp = lambda t, S: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3).clone().requires_grad_(True)
    qr, kr, vr = sp(q, Sq), sp(k, Sk), sp(v, Sk)
    ref, _ = _attn_ref(qr, kr, vr, scale)
    refl = ref.permute(0, 2, 1, 3).reshape(B, Sq, C)
    assert relerr(o, refl) < 4e-3
    # backward
    go = rnd(B, Sq, C, dev=dev, seed=4).half()
    ref.backward(go.float().reshape(B, Sq, H, d).permute(0, 2, 1, 3))
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Sq, device=dev)
    ops.attn_bwd(q, k, v, o, go, lse, delta, gq, gk, gv, B, H, Sq, Sk, d, scale)
    un = lambda t, S: t.permute(0, 2, 1, 3).reshape(B, S, C)
    assert relerr(gq, un(qr.grad, Sq)) < 1e-2
    assert relerr(gk, un(kr.grad, Sk)) < 1e-2
    assert relerr(gv, un(vr.grad, Sk)) < 1e-2


@pytest.mark.parametrize("B,H,S,d", [(2, 8, 4096, 40), (2, 8, 1024, 80), (4, 8, 256, 160), (1, 3, 200, 40)])
def test_attn_bwd_key_gradients_for_the_visual_rows_only(dev, B, H, S, d):
    """lgd_attn_bwd_keys_f16 (ABI v10): the GLIGEN fuser's attention over [S visual ; 30 grounding] keys with dK / dV wanted for
    the S visual keys only (attention.py:43-53: the grounding rows are constants of a run) — dQ and the first S rows of dK / dV
    bit-identical to the full backward, rows >= S untouched."""
    Sk, C, scale = S + 30, H * d, d ** -0.5
    q = rnd(B, S, C, dev=dev, seed=1).half()
    k = rnd(B, Sk, C, dev=dev, seed=2).half()
    v = rnd(B, Sk, C, dev=dev, seed=3).half()
    go = rnd(B, S, C, dev=dev, seed=4).half()
    o = torch.empty(B, S, C, device=dev, dtype=H16)
    lse = torch.empty(B, H, S, device=dev)
    ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, scale, lse=lse)
    outs = []
    for skg in (None, S):
        gq, gk, gv = torch.full_like(q, 7.0), torch.full_like(k, 7.0), torch.full_like(v, 7.0)
        delta = torch.empty(B, H, S, device=dev)
        ops.attn_bwd(q, k, v, o, go, lse, delta, gq, gk, gv, B, H, S, Sk, d, scale, sk_grad=skg)
        outs.append((gq, gk, gv))
    (gq0, gk0, gv0), (gq1, gk1, gv1) = outs
    assert torch.equal(gq0, gq1)
    assert torch.equal(gk0[:, :S], gk1[:, :S]) and torch.equal(gv0[:, :S], gv1[:, :S])
    assert bool((gk1[:, S:] == 7.0).all()) and bool((gv1[:, S:] == 7.0).all())
    assert not bool((gk0[:, S:] == 7.0).all())


@pytest.mark.parametrize("B,H,Sq,Sk,d", [
    (2, 8, 4096, 4096, 40), (1, 8, 300, 333, 40), (2, 8, 1024, 1054, 80), (1, 3, 77, 64, 40), (1, 2, 256, 1, 80),
    (1, 8, 4096, 4126, 40), (1, 4, 129, 257, 24), (1, 2, 500, 190, 56), (1, 2, 64, 700, 88),
])
def test_attn_self32_kernel_every_size(dev, B, H, Sq, Sk, d):
    """The round-3 self-attention forward (32x32x16 MFMA, in-wave pipelining; attention_processor.py:338-363 semantics)
    forced for every problem size (lgd_set_option("attn32", 2)): ragged query / key counts (masked through the spare head-dim slot),
    single-tile and single-key cases, every head dim it serves (d + 2 <= 48 or 96), output and log-sum-exp vs fp32
    torch, and bit-for-bit determinism; with spiked keys the lazily raised reference must rescale."""
    ops.set_option("attn32", 2)
    try:
        _attn_self32_case(dev, B, H, Sq, Sk, d)
    finally:
        ops.set_option("attn32", 1)


def _attn_self32_case(dev, B, H, Sq, Sk, d):
    C = H * d
    scale = d ** -0.5
    q = rnd(B, Sq, C, dev=dev, seed=1).half()
    k = rnd(B, Sk, C, dev=dev, seed=2).half()
    v = rnd(B, Sk, C, dev=dev, seed=3).half()
    sp = lambda t, S: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    for spike in (False, True):
        if spike:
            if Sk < 8:
                continue
            for qi, ki in [(5, Sk - 3), (Sq // 2 + 1, min(Sk - 1, Sk // 2 + 70)), (Sq - 1, min(Sk - 1, 200))]:
                k[:, ki] = q[:, qi] * 6.0
        o = torch.empty(B, Sq, C, device=dev, dtype=H16)
        lse = torch.empty(B, H, Sq, device=dev)
        ops.attn_fwd(q, k, v, o, B, H, Sq, Sk, d, scale, lse=lse)
        o2 = torch.empty_like(o)
        ops.attn_fwd(q, k, v, o2, B, H, Sq, Sk, d, scale)
        ref, _ = _attn_ref(sp(q, Sq), sp(k, Sk), sp(v, Sk), scale)
        e = relerr(o, ref.permute(0, 2, 1, 3).reshape(B, Sq, C))
        lse_ref = torch.logsumexp(torch.einsum("bhqd,bhkd->bhqk", sp(q, Sq), sp(k, Sk)) * scale, dim=-1) * 1.4426950408889634
        el = float((lse - lse_ref).abs().max())
        print(f"attn32 B{B} H{H} {Sq}x{Sk} d{d} spike={spike}: relerr {e:.2e}, lse abs err {el:.2e}")
        assert torch.isfinite(o).all() and torch.equal(o, o2)
        assert e < 4e-3 and el < 2e-2
    ops.set_option("attn32", 0)                        # and the 16x16x32 kernel it replaces agrees with it
    o3 = torch.empty_like(o)
    ops.attn_fwd(q, k, v, o3, B, H, Sq, Sk, d, scale)
    assert relerr(o3, o) < 4e-3


def test_attn_qkv_fused_view(dev):
    """q/k/v taken as column views of one fused [B,S,3C] projection output."""
    B, H, S, d = 2, 8, 256, 40
    C = H * d
    qkv = rnd(B, S, 3 * C, dev=dev, seed=1).half()
    o = torch.empty(B, S, C, device=dev, dtype=H16)
    view = (3 * C, S * 3 * C)
    ops.attn_fwd(qkv, qkv[:, :, C:], qkv[:, :, 2 * C:], o, B, H, S, S, d, d ** -0.5,
                 q_view=view, k_view=view, v_view=view)
    sp = lambda t: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    ref, _ = _attn_ref(sp(qkv[..., :C]), sp(qkv[..., C:2 * C]), sp(qkv[..., 2 * C:]), d ** -0.5)
    assert relerr(o, ref.permute(0, 2, 1, 3).reshape(B, S, C)) < 4e-3


@pytest.mark.parametrize("B,H,Sq,d,tok,cond_only", [
    (2, 8, 4096, 40, -1, False), (2, 8, 256, 160, 5, True), (1, 8, 64, 160, -1, False),
    (2, 8, 1024, 80, -1, True), (1, 8, 256, 8, 3, False),
])
def test_cross_attn_maps(dev, B, H, Sq, d, tok, cond_only):
    Sk, C = 77, H * d
    scale = d ** -0.5
    q = (rnd(B, Sq, C, dev=dev, seed=1) * 2).half()
    k = rnd(B, Sk, C, dev=dev, seed=2).half()
    v = rnd(B, Sk, C, dev=dev, seed=3).half()
    o = torch.empty(B, Sq, C, device=dev, dtype=H16)
    Bp = B // 2 if cond_only else B
    Tp = 1 if tok >= 0 else Sk
    probs = torch.zeros(Bp, H, Sq, Tp, device=dev)
    ops.cross_attn_fwd(q, k, v, o, B, H, Sq, Sk, d, scale, probs=probs, tok=tok, cond_only=cond_only)
    sp = lambda t, S: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    ref, p = _attn_ref(sp(q, Sq), sp(k, Sk), sp(v, Sk), scale)
    assert relerr(o, ref.permute(0, 2, 1, 3).reshape(B, Sq, C)) < 4e-3
    if tok >= 0:
        p = p[..., tok:tok + 1]
    if cond_only:
        p = p[B // 2:]
    assert relerr(probs, p) < 2e-3
    # no-map variant must agree with the map variant
    o2 = torch.empty_like(o)
    ops.cross_attn_fwd(q, k, v, o2, B, H, Sq, Sk, d, scale)
    assert relerr(o2, o) < 2e-3


@pytest.mark.parametrize("B,H,Sq,d,with_go,with_gp", [
    (1, 8, 256, 160, True, True), (1, 8, 64, 160, False, True), (2, 8, 1024, 80, True, False),
    (1, 8, 100, 8, True, True), (2, 8, 4096, 40, True, False), (1, 8, 300, 40, True, True),
])
def test_cross_attn_bwd(dev, B, H, Sq, d, with_go, with_gp):
    Sk, C = 77, H * d
    scale = d ** -0.5
    q = rnd(B, Sq, C, dev=dev, seed=1).half()
    k = rnd(B, Sk, C, dev=dev, seed=2).half()
    v = rnd(B, Sk, C, dev=dev, seed=3).half()
    sp = lambda t, S: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    qr = sp(q, Sq).clone().requires_grad_(True)
    ref, p = _attn_ref(qr, sp(k, Sk), sp(v, Sk), scale)
    go = rnd(B, Sq, C, dev=dev, seed=4).half() if with_go else None
    gp = rnd(B, H, Sq, Sk, dev=dev, seed=5) if with_gp else None
    loss = 0
    if with_go:
        loss = loss + (ref * sp(go, Sq)).sum()
    if with_gp:
        loss = loss + (p * gp).sum()
    loss.backward()
    gq = torch.empty_like(q)
    ops.cross_attn_bwd(q, k, v, go, gp, gq, B, H, Sq, Sk, d, scale)
    assert relerr(gq, qr.grad.permute(0, 2, 1, 3).reshape(B, Sq, C)) < 5e-3


def test_geglu_bwd_and_elementwise(dev):
    rows, n = 300, 1280
    w = rnd(rows, 2 * n, dev=dev, seed=1).half()  # natural layout [v | g]
    hp, _ = None, None
    # pack natural -> 16-blocks along columns
    idx = []
    for j in range(n // 16):
        idx += list(range(16 * j, 16 * j + 16)) + list(range(n + 16 * j, n + 16 * j + 16))
    idx = torch.tensor(idx, device=dev)
    hpk = w[:, idx].contiguous()
    gy = rnd(rows, n, dev=dev, seed=2).half()
    gh = ops.geglu_bwd(hpk, gy)
    wr = w.float().clone().requires_grad_(True)
    vv, gg = wr.chunk(2, dim=-1)
    (vv * F.gelu(gg)).backward(gy.float())
    assert relerr(gh, wr.grad[:, idx]) < 3e-3
    a, b = rnd(1000, 64, dev=dev, seed=3).half(), rnd(1000, 64, dev=dev, seed=4).half()
    assert relerr(ops.add(a, b), a.float() + b.float()) < 1e-3
    assert relerr(ops.scale(a, 0.25), a.float() * 0.25) < 1e-3
    B, H, W, C = 2, 8, 8, 64
    g = rnd(B * 4 * H * W, C, dev=dev, seed=5).half()
    gx = ops.upsample2x_bwd(g, B, H, W, C)
    ref = g.float().reshape(B, H, 2, W, 2, C).sum(dim=(2, 4)).reshape(-1, C)
    assert relerr(gx, ref) < 2e-3


def test_cfg_ddim_step(dev):
    B, C, L, T = 2, 4, 64, 5
    eps = rnd(2 * B, C, L, L, dev=dev, seed=1)
    x = rnd(B, C, L, L, dev=dev, seed=2)
    table = torch.tensor([[0.5 + 0.05 * i, 0.6 + 0.05 * i, 7.5, 0.0] for i in range(T)], device=dev)
    ref_lat = rnd(T + 1, B, C, L, L, dev=dev, seed=3)
    mask = (rnd(B, L * L, dev=dev, seed=4) > 0).float()
    hist = torch.zeros(T + 1, B, C, L, L, device=dev)
    for step, frozen_steps, vpred in [(1, 3, 0.0), (4, 3, 0.0), (2, 0, 1.0)]:
        table[:, 3] = vpred
        idx = torch.tensor([step, frozen_steps], device=dev, dtype=torch.int32)
        out = torch.empty_like(x)
        ops.cfg_ddim_step(eps, x, out, table, idx, frozen_ref=ref_lat, mask=mask, hist=hist)
        a_t, a_p = table[step, 0], table[step, 1]
        e = eps[:B] + 7.5 * (eps[B:] - eps[:B])
        if vpred:
            x0 = a_t.sqrt() * x - (1 - a_t).sqrt() * e
            e = a_t.sqrt() * e + (1 - a_t).sqrt() * x
        else:
            x0 = (x - (1 - a_t).sqrt() * e) / a_t.sqrt()
        xn = a_p.sqrt() * x0 + (1 - a_p).sqrt() * e
        if step < frozen_steps:
            m = mask.reshape(B, 1, L, L)
            xn = ref_lat[step + 1] * m + xn * (1 - m)
        assert relerr(out, xn) < 1e-5
        assert relerr(hist[step + 1], xn) < 1e-5
    g = rnd(B, C, L, L, dev=dev, seed=9)
    x2 = x.clone()
    ops.axpy(g, x2, table, torch.tensor([2], device=dev, dtype=torch.int32), 1)
    assert relerr(x2, x - table[2, 1] * g) < 1e-6
    out = torch.empty(4, device=dev)
    ops.select_row(table, torch.tensor([3], device=dev, dtype=torch.int32), out)
    assert torch.equal(out, table[3])


@pytest.mark.parametrize("pipe", [1, 0])
@pytest.mark.parametrize("B,H,Sq,Sk,kw", [
    (2, 8, 4096, 4096, {}), (2, 8, 4096, 4126, {}), (1, 2, 300, 77, {}), (1, 3, 257, 64, {}), (2, 2, 64, 1, {}),
    (1, 2, 1000, 129, {}), (2, 8, 4096, 4096, dict(spike=True)), (2, 4, 2048, 2111, dict(spike=True)),
    (2, 8, 4096, 4096, dict(scale_q=6.0)), (1, 1, 31, 200, {}), (1, 2, 256, 192, {}), (1, 2, 512, 320, {}),
])
def test_attn_w4_kernel_every_size(dev, pipe, B, H, Sq, Sk, kw):
    """The round-4 d = 40 self-attention forward (csrc/attn_w4.hip: attention_processor.py:338-363 semantics), forced for
    every problem size (lgd_set_option("attn_w4", 2)), in both variants (one wave per SIMD with the in-wave software
    pipeline / two waves per SIMD): ragged query and key counts (padded keys dropped through the DMA'd constant lines),
    1 .. 65 key tiles (prologue-only, one pipelined tile, steady-state loop, generic tail), spiked keys and large
    logits (the lazily raised reference must rescale), output and log-sum-exp vs fp32 torch on EVERY (image, head), no
    NaN anywhere, bit-for-bit determinism."""
    d = 40
    C = H * d
    g = torch.Generator().manual_seed(7)
    q = (torch.randn(B, Sq, C, generator=g) * kw.get("scale_q", 1.0)).to(dev).half()
    k = torch.randn(B, Sk, C, generator=g).to(dev).half()
    v = torch.randn(B, Sk, C, generator=g).to(dev).half()
    if kw.get("spike"):
        for j in (Sk // 2 + 3, Sk - 5):
            k[:, j] = q[:, (j * 7) % Sq] * 6.0
    ops.set_option("attn_w4", 2)
    ops.set_option("attn_w4_pipe", pipe)
    try:
        outs = []
        for _ in range(2):
            o = torch.full((B, Sq, C), float("nan"), device=dev, dtype=H16)
            lse = torch.full((B, H, Sq), float("nan"), device=dev)
            ops.attn_fwd(q, k, v, o, B, H, Sq, Sk, d, d ** -0.5, lse=lse)
            torch.cuda.synchronize()
            outs.append((o, lse))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        o, lse = outs[0]
        assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all())
        sp = lambda t, S: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
        logits = torch.einsum("bhqd,bhkd->bhqk", sp(q, Sq), sp(k, Sk)) * d ** -0.5
        ref = torch.einsum("bhqk,bhkd->bhqd", logits.softmax(-1), sp(v, Sk)).permute(0, 2, 1, 3).reshape(B, Sq, C)
        gate(f"[attn_w4 pipe={pipe} B{B} H{H} {Sq}x{Sk} {kw}] output", relerr(o, ref), 4e-3)
        gate(f"[attn_w4 pipe={pipe} B{B} H{H} {Sq}x{Sk} {kw}] log-sum-exp (log2 domain, abs)",
             float((lse - torch.logsumexp(logits, -1) * 1.4426950408889634).abs().max()), 2e-2)
    finally:
        ops.set_option("attn_w4", 1)
        ops.set_option("attn_w4_pipe", 1)
