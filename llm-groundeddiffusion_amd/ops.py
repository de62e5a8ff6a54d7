"""Thin host wrappers over the C ABI: torch tensors in, raw device pointers across the boundary.

torch is used for device memory and the current HIP stream only.  Every function enqueues kernels on
`torch.cuda.current_stream()` and returns immediately (graph-capturable).  No fallbacks: a failing
call raises RuntimeError (the error convention of the reference's plugin boundary,
generate.py:391-396).
"""
import contextlib
import ctypes as C

import torch

from . import _lib
from ._lib import EPI_GEGLU, EPI_OUT_F32, EPI_RES_F32, EPI_ROWNORM, LgdGemmDesc

F16, F32 = torch.float16, torch.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _call(name, *args):
    lib = _lib.load()
    rc = getattr(lib, name)(*args)
    _lib.check(rc, name)


def set_option(name: str, value: int):
    """Kernel-variant switch of the library (lgd_set_option): e.g. set_option("attn32", 0 | 1 | 2)."""
    _call("lgd_set_option", name.encode(), int(value))


# ---------------------------------------------------------------------------------------------
# GEMM / conv
# ---------------------------------------------------------------------------------------------
import threading as _threading
_TLS = _threading.local()   # split-K workspace of ad-hoc calls: one per host thread (= one per HIP stream lane, lanes.py)
WS_FLOATS = 1 << 26


def workspace(device) -> torch.Tensor:
    """fp32 split-K scratch for calls that bring none.  Launches of ONE host thread run back to back on its current
    stream and may share it; two threads driving two streams (lanes.LanePool) must not, hence thread-local.  Engines
    own their own (UNetEngine.workspace): their descriptors are baked into plans and hipGraphs."""
    dev = torch.device(device)
    ws = getattr(_TLS, "ws", None)
    if ws is None or ws.device != dev:
        ws = _TLS.ws = torch.empty(WS_FLOATS, device=dev, dtype=F32)
    return ws


N_COUNTERS = 1 << 16
import os as _os
# In-launch split-K combine ("last arriver reduces", LgdGemmDesc.cnt): implemented, bit-identical, and MEASURED SLOWER on
# MI355X than the second launch it replaces — every workgroup's agent-scope release is a write-back of its XCD's L2
# (M = 256, K = 11520, 12 splits: 57 us vs 26 us; default bench 1.21 vs 1.28 images/s) — so it stays off.
SPLITK_IN_LAUNCH = _os.environ.get("LGD_SPLITK_IN_LAUNCH", "0") == "1"


def splitk_counters(device):
    # one buffer per host thread = per lane = per stream: the header allows one counter buffer to serve the GEMMs of ONE
    # stream only (concurrent launches of one shape on two streams would share ticket slots).  Thread-local, like the
    # split-K scratch: the buffer goes away with its lane thread instead of accumulating under dead thread ids
    dev = torch.device(device)
    cnt = getattr(_TLS, "cnt", None)
    if cnt is None or cnt.device != dev:
        cnt = _TLS.cnt = torch.zeros(N_COUNTERS, device=dev, dtype=torch.int32)
    return cnt


def choose_splits(M, N, K, batches=1):
    """Split-K heuristic (used when the tuning table has no entry): spread small-M problems
    (8x8 / 16x16 levels) over all 256 CUs."""
    wgs = ((M + 127) // 128) * ((N + 63) // 64) * batches
    if wgs >= 256:
        return 1
    ktiles = K // 64
    s = min((512 + wgs - 1) // wgs, max(1, ktiles // 4), 16)
    return max(1, s)


def gemm_desc(a0, w, c, M, N, K, *, a1=None, lda0=None, lda1=0, c0=None, c1=0, taps=1,
              hin=0, win=0, hout=0, wout=0, stride=1, ups=0, ldw=None, bias=None, bias2=None,
              res=None, ldr=0, alpha=1.0, epi=0, ldc=None, splits=None, ws=None, tile=0,
              nb_o=1, nb_i=1, a_bs=(0, 0), w_bs=(0, 0), c_bs=(0, 0), r_bs=(0, 0), rowstat=None, colsum=None):
    """Builds an LgdGemmDesc from raw pointers (ints) or tensors.  splits=None / tile=0: taken from
    the measured tuning table (tuning_gfx950.json) when the shape is listed, else from heuristics."""
    d = LgdGemmDesc()
    ptr = lambda t: (t.data_ptr() if torch.is_tensor(t) else (t or 0))
    d.a0, d.a1 = ptr(a0), ptr(a1)
    cin = K // taps
    d.c0 = cin - c1 if c0 is None else c0
    d.c1 = c1
    d.lda0 = d.c0 if lda0 is None else lda0
    d.lda1 = lda1 if lda1 else max(c1, 0)
    d.taps, d.hin, d.win, d.hout, d.wout, d.stride, d.ups = taps, hin, win, hout, wout, stride, ups
    d.w = ptr(w)
    d.ldw = K if ldw is None else ldw
    d.M, d.N, d.K = M, N, K
    d.nb_o, d.nb_i = nb_o, nb_i
    d.a_bs_o, d.a_bs_i = a_bs
    d.w_bs_o, d.w_bs_i = w_bs
    d.c_bs_o, d.c_bs_i = c_bs
    d.r_bs_o, d.r_bs_i = r_bs
    d.bias, d.bias2 = ptr(bias), ptr(bias2)
    d.res, d.ldr = ptr(res), ldr
    if rowstat is not None:                              # rows arrive raw, w / bias / colsum are the folded set (weightstore)
        epi |= EPI_ROWNORM
    d.rowstat, d.colsum = ptr(rowstat), ptr(colsum)
    d.alpha, d.epi = alpha, epi
    d.c = ptr(c)
    n_out = N // 2 if (epi & EPI_GEGLU) else N
    d.ldc = n_out if ldc is None else ldc
    ent = tuning_table().get(shape_key(d)) if (splits is None or not tile) else None
    if splits is None:
        splits = ent["splits"] if ent else choose_splits(M, N, K, nb_o * nb_i)
    if ent and not _table_tile_applies(ent["tile"], d, splits, epi, n_out):
        ent = None                                        # same M/N/K key, but an epilogue form that tile does not have
    if not tile:
        tile = ent["tile"] if (ent and ent["splits"] == splits) else choose_tile(M, N, nb_o * nb_i * max(splits, 1), bool(epi & EPI_GEGLU), K,
                                                                             pipe_ok=(K % 64 == 0 and d.c0 % 64 == 0 and d.c1 % 64 == 0))
    if splits > 1 and ws is None:
        need = splits * M * N * nb_o * nb_i
        ws = workspace(c.device if torch.is_tensor(c) else "cuda")
        if need > ws.numel():
            raise RuntimeError(f"split-K workspace too small for {splits}x{M}x{N}")
    d.splits, d.ws = splits, ptr(ws)
    d.tile = tile
    log = getattr(_TUNING_TLS, "log", None)
    if log is not None:                                  # tests: which table entries a plan was built from
        log.append((shape_key(d), tile, splits))
    d.cnt = 0
    if splits > 1 and SPLITK_IN_LAUNCH and torch.is_tensor(c) and c.is_cuda and tile not in PHASE_TILES:   # phase tiles: reduce launch only
        # the smallest tile of the library is 32 rows x 64 columns: an upper bound of the launch's output tiles
        if nb_o * nb_i * ((M + 31) // 32) * ((N + 63) // 64) <= N_COUNTERS:
            d.cnt = splitk_counters(c.device).data_ptr()
    return d


def _table_tile_applies(tile, d, splits, epi, n_out) -> bool:
    """The tuning table is keyed by shape_key (M, N, K, gather, GEGLU flag), not by the whole epilogue form.  The two-stage
    ring tiles of round 4 take plain single-source contractions only, and the 256 x 256 one (44) leaves through the LDS
    epilogue alone: one split, fp16 rows of N % 8 = 0 with an 8-aligned leading dimension and a 16-byte aligned base, no
    fp32 output / residual, not GEGLU together with a residual (launch_gemm_pipe returns LGD_ERR_ARG otherwise).  A caller
    with the same M/N/K but such an epilogue must get the heuristic tile, not a RuntimeError."""
    if tile in PHASE_TILES:
        # round 6, phase-split 256-row tiles: plain single-source contractions and 3x3 stride-1 same-size convolutions;
        # one split through the LDS epilogue (as tile 44), several splits through fp32 partials; no GEGLU on 256 x 320
        if d.c1 != 0 or d.nb_o * d.nb_i != 1 or (d.taps == 9 and (d.stride != 1 or d.ups != 0 or d.hin != d.hout or d.win != d.wout)):
            return False
        if (epi & EPI_GEGLU) and tile == 47:
            return False
        if splits != 1:
            return True
        if (epi & (EPI_OUT_F32 | EPI_RES_F32)) or ((epi & EPI_GEGLU) and d.res):
            return False
        return n_out % 8 == 0 and d.ldc % 8 == 0 and d.c % 16 == 0
    if tile not in (44, 45):
        return True
    if d.taps != 1 or d.c1 != 0:
        return False
    if tile == 45:
        return True
    if splits != 1 or (epi & (EPI_OUT_F32 | EPI_RES_F32)) or ((epi & EPI_GEGLU) and d.res):
        return False
    return n_out % 8 == 0 and d.ldc % 8 == 0 and d.c % 16 == 0


_TILE_DIMS = {1: (4, 4), 2: (4, 2), 3: (2, 4), 4: (2, 2), 5: (1, 4), 6: (4, 5), 7: (2, 5)}
TILE_NAMES = {t: f"gemm_kernel<{mi},{ni}> {32 * mi}x{32 * ni}" for t, (mi, ni) in _TILE_DIMS.items()}
TILE_NAMES.update({16 + t: f"gemm_dma_kernel<{mi},{ni},2> {32 * mi}x{32 * ni}" for t, (mi, ni) in _TILE_DIMS.items()})
TILE_NAMES.update({25: "gemm_dma_kernel<4,10,4> 256x320", 26: "gemm_dma_kernel<4,4,4> 256x128"})
# 8-wave pipelined main loop: code -> (MI, NI, stages); tile = 64*MI x 32*NI
_PIPE_DIMS = {33: (4, 5, 3), 34: (4, 4, 3), 35: (4, 2, 4), 37: (2, 5, 4), 38: (2, 4, 4), 39: (2, 2, 5), 40: (1, 5, 5),
              41: (1, 4, 5), 42: (1, 2, 6)}
TILE_NAMES.update({t: f"gemm_pipe_kernel<{mi},{ni},4,2,{ns}> {64 * mi}x{32 * ni}" for t, (mi, ni, ns) in _PIPE_DIMS.items()})
# round 4: 256 x 256, two-stage ring (plain single-source contractions only), eight waves of 64 x 128
TILE_NAMES.update({44: "gemm_pipe_kernel<4,8,4,2,2> 256x256", 45: "gemm_pipe_kernel<2,4,4,2,2> 128x128 x2/CU"})
# round 6: phase-split main loop (two wave groups one interval apart, four phases per K tile), 2 x 4 waves of 128 x (64 | 80)
PHASE_TILES = {46: (256, 256), 47: (256, 320)}
TILE_NAMES.update({46: "gemm_phase_kernel<8,4> 256x256", 47: "gemm_phase_kernel<8,5> 256x320"})


def choose_tile(M, N, batches=1, geglu=False, K=64, pipe_ok=False):
    """Tile heuristic for shapes the tuning table does not list (same rule as the library's auto mode,
    done here so the choice is known to the profiler): the largest tile that still yields enough
    workgroups for 256 CUs; 160-wide tiles when they divide N (every UNet width is 320 k).  Codes
    17..23 select the LDS-DMA main loop (the library falls back to register staging if K % 64)."""
    wgs = lambda bm, bn: batches * ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    if M <= 32:
        return 21
    if pipe_ok:
        # shapes outside the measured table (VAE decoder, SAM, batch sizes the tuner did not visit): the 8-wave pipelined
        # tiles, largest first, as soon as they fill the chip once — what the table picks for 2/3 of the UNet's shapes
        if not geglu and N % 160 == 0 and wgs(256, 160) >= 256:
            return 33
        if N % 128 == 0 and wgs(256, 128) >= 256:
            return 34
        if not geglu and N % 160 == 0 and wgs(128, 160) >= 256:
            return 37
        if N % 128 == 0 and wgs(128, 128) >= 256:
            return 38
    if not geglu and N % 160 == 0:
        if wgs(128, 160) >= 384:
            return 22
        if wgs(64, 160) >= 256:
            return 23
    if wgs(128, 128) >= 384 and N % 128 == 0:
        return 17
    if wgs(64, 128) >= 256 and N % 128 == 0:
        return 19
    return 20


def shape_key(d) -> str:
    """Identity of a GEMM launch for the tuning table (everything that changes its cost)."""
    return (f"M{d.M}_N{d.N}_K{d.K}_t{d.taps}_c{d.c0}+{d.c1}_h{d.hin}x{d.hout}_s{d.stride}_u{d.ups}"
            f"_e{d.epi & 1}_b{d.nb_o * d.nb_i}")


_TUNING = {}
# "latency": tile / split-K per shape chosen for the shortest isolated launch (one launch sequence owns the GPU);
# "throughput": chosen for the lowest cost when several sequences share the GPU (lanes.LanePool; the same launch on
# 4 streams at once, tools/tune_gemm.py LGD_TUNE_STREAMS=4): fewer split-K slabs, larger tiles — a launch may leave CUs
# idle, another lane fills them.  Plans read the table when they are BUILT.
TUNING_MODE = _os.environ.get("LGD_TUNING_MODE", "latency")
# "heuristic": no table at all — choose_tile / choose_splits for every shape: a launch configuration that does not move
# when the tables are re-tuned (the pinned reference point of the full-width gradient gate, tests/test_bench_path_gpu.py)
_TUNING_FILES = {"latency": ("tuning_gfx950.json",), "throughput": ("tuning_gfx950.json", "tuning_gfx950_lanes.json"),
                 "heuristic": ()}
_TUNING_TLS = _threading.local()


def set_tuning_mode(mode: str):
    """Process-wide DEFAULT (tools / A-B runs).  Engines and lanes do not use it: a lane engine carries its own
    `tuning_mode` and a lane thread selects the table with `tuning(...)` for the launches it describes itself."""
    global TUNING_MODE
    if mode not in _TUNING_FILES:
        raise ValueError(f"tuning mode {mode!r}: expected one of {sorted(_TUNING_FILES)}")
    if _os.environ.get("LGD_TUNING_MODE"):       # an explicit environment choice wins (A/B measurements)
        return
    TUNING_MODE = mode


def current_tuning_mode() -> str:
    if _os.environ.get("LGD_TUNING_MODE"):
        return TUNING_MODE
    return getattr(_TUNING_TLS, "mode", None) or TUNING_MODE


@contextlib.contextmanager
def tuning(mode):
    """Table selection for the GEMM descriptors built by THIS thread inside the block (None = leave as is): plan
    construction of an engine, the body of a lane thread.  Nothing process-wide changes."""
    if mode is not None and mode not in _TUNING_FILES:
        raise ValueError(f"tuning mode {mode!r}: expected one of {sorted(_TUNING_FILES)}")
    prev = getattr(_TUNING_TLS, "mode", None)
    if mode is not None:
        _TUNING_TLS.mode = mode
    try:
        yield
    finally:
        _TUNING_TLS.mode = prev


@contextlib.contextmanager
def desc_log():
    """Collects (shape key, tile, splits) of every GEMM descriptor this thread builds inside the block."""
    prev, _TUNING_TLS.log = getattr(_TUNING_TLS, "log", None), []
    try:
        yield _TUNING_TLS.log
    finally:
        _TUNING_TLS.log = prev


def tuning_table():
    mode = current_tuning_mode()
    if mode not in _TUNING:
        import json
        tab = {}
        for fname in _TUNING_FILES[mode]:      # later files override earlier ones shape by shape
            path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), fname)
            if _os.path.exists(path):
                tab.update(json.load(open(path)))
        _TUNING[mode] = tab
    return _TUNING[mode]


class LaunchProfiler:
    """Samples kernel durations with HIP events on the launch stream (torch.cuda.Event records on
    torch's current stream, which is the stream every lgd_* call is enqueued on)."""

    def __init__(self, max_records=20000):
        self.recs = []
        self.max = max_records
        self.enabled = True

    def wrap(self, name, flops, nbytes, fn, tag=None, shape=None):
        if not self.enabled or len(self.recs) >= self.max:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.recs.append((name, flops, nbytes, e0, e1, tag, shape))
        return r

    def summary(self, by_tag=False, by_shape=False):
        """Per kernel name (or per caller tag, e.g. "attn_path"; or per launch shape "kernel | shape key"): ms,
        algorithmic flops / bytes, launches."""
        torch.cuda.synchronize()
        agg = {}
        for name, fl, nb, e0, e1, tag, shape in self.recs:
            if by_tag and tag is None:
                continue
            key = tag if by_tag else (f"{name} | {shape}" if by_shape else name)
            a = agg.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += fl
            a["bytes"] += nb
            a["n"] += 1
        return agg


PROFILER = None


def _prof(name, nbytes, fn, *, flops=0.0, tag=None, shape=None):
    """Runs `fn` (one kernel launch, or the few launches of one C-ABI entry point); under a LaunchProfiler the launch is
    bracketed by HIP events and booked under `name` with its ALGORITHMIC bytes / flops.  Every entry point a UNet plan
    or a sampler step enqueues goes through here or through gemm_launch / attn_fwd, so that the profiler's kernels sum
    to the launch sequence's time (bench.py `roofline.all_kernels`)."""
    if PROFILER is not None:
        return PROFILER.wrap(name, flops, nbytes, fn, tag, shape)
    return fn()


def copy_(dst, src):
    """dst.copy_(src) on the current stream, visible to the launch profiler (plan-internal copies: CFG duplication of the
    latents, first-writer gradient copies of residual branches, captured-map slices)."""
    return _prof("copy", 2.0 * dst.numel() * dst.element_size(), lambda: dst.copy_(src))


def zero_(t):
    return _prof("fill", float(t.numel() * t.element_size()), lambda: t.zero_())


def gemm_launch(desc, tag=None, flops=None):
    """`flops`: algorithmic work when it differs from 2*M*N*K of the launch (conv_in multiplies a zero / remainder-padded K)."""
    if PROFILER is not None:
        nb = desc.nb_o * desc.nb_i
        if flops is None:
            flops = 2.0 * desc.M * desc.N * desc.K * nb
        nbytes = 2.0 * nb * (desc.M * desc.K / max(desc.taps, 1) + desc.N * desc.K + desc.M * desc.N)
        return PROFILER.wrap(TILE_NAMES.get(desc.tile, "gemm"), flops, nbytes,
                             lambda: _call("lgd_gemm_f16", C.byref(desc), _stream()), tag,
                             shape=f"{shape_key(desc)} splits={desc.splits}")
    _call("lgd_gemm_f16", C.byref(desc), _stream())


def linear(x, w, bias=None, res=None, out=None, *, alpha=1.0, geglu=False, out_f32=False,
           splits=None, ws=None, tile=0, bias2=None, rowstat=None, colsum=None):
    """y = x @ w.T (+bias) ... ; x [M,K] fp16 contiguous, w [N,K] fp16."""
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), device=x.device, dtype=F32 if out_f32 else F16)
    epi = (EPI_GEGLU if geglu else 0) | (EPI_OUT_F32 if out_f32 else 0)
    if res is not None and res.dtype == F32:
        epi |= EPI_RES_F32
    d = gemm_desc(x, w, out, M, N, K, bias=bias, bias2=bias2, res=res,
                  ldr=(res.stride(0) if res is not None else 0), alpha=alpha, epi=epi,
                  splits=splits, ws=ws, tile=tile, lda0=x.stride(0), ldw=w.stride(0),
                  ldc=out.stride(0), rowstat=rowstat, colsum=colsum)
    gemm_launch(d)
    return out


def conv3x3(x, w, B, H, W, *, x1=None, bias=None, bias2=None, res=None, stride=1, ups=False,
            out=None, splits=None, ws=None, tile=0, alpha=1.0):
    """3x3 conv, pad 1, channels-last.  x [B*H*W, C0] (stored map; with ups the logical input is
    2H x 2W), optional x1 [B*H*W, C1] concatenated on channels; w [Cout, 9*(C0+C1)]."""
    c0 = x.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    Cout = w.shape[0]
    K = 9 * (c0 + c1)
    assert w.shape[1] == K
    if ups:
        Hl, Wl = 2 * H, 2 * W
    else:
        Hl, Wl = H, W
    Ho, Wo = (Hl - 1) // stride + 1, (Wl - 1) // stride + 1
    M = B * Ho * Wo
    if out is None:
        out = torch.empty((M, Cout), device=x.device, dtype=F16)
    d = gemm_desc(x, w, out, M, Cout, K, a1=x1, c0=c0, c1=c1, lda0=x.stride(0),
                  lda1=(x1.stride(0) if x1 is not None else 0), taps=9, hin=H, win=W, hout=Ho,
                  wout=Wo, stride=stride, ups=int(ups), bias=bias, bias2=bias2, res=res,
                  ldr=(res.stride(0) if res is not None else 0), splits=splits, ws=ws, tile=tile,
                  alpha=alpha, ldc=out.stride(0))
    gemm_launch(d)
    return out


def conv_in(x_nchw, w, bias, out=None):
    B, Cin, L, _ = x_nchw.shape
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((B * L * L, Cout), device=x_nchw.device, dtype=F16)
    _call("lgd_conv_in_f16", _p(x_nchw), _p(w), _p(bias), _p(out), B, Cin, L, Cout, _stream())
    return out


def nchw_to_nhwc8(x_nchw, out=None):
    """(B, C<=8, H, W) fp32 -> [B*H*W, 8] fp16, zero channels behind C."""
    B, C_, H, W = x_nchw.shape
    if out is None:
        out = torch.empty((B * H * W, 8), device=x_nchw.device, dtype=F16)
    _prof("nchw_to_nhwc8_kernel", 4.0 * x_nchw.numel() + 2.0 * out.numel(),
          lambda: _call("lgd_nchw_to_nhwc8_f16", _p(x_nchw), _p(out), B, C_, H * W, _stream()))
    return out


def conv_out(x, w, bias, B, L, out=None, out_scale=1.0):
    Cin = x.shape[1]
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((B, Cout, L, L), device=x.device, dtype=F32)
    _prof("conv_out_kernel", 2.0 * x.numel() + 4.0 * out.numel(),
          lambda: _call("lgd_conv_out_f16", _p(x), _p(w), _p(bias), _p(out), B, Cin, L, Cout, float(out_scale), _stream()),
          flops=2.0 * B * L * L * Cout * 9 * Cin)
    return out


# ---------------------------------------------------------------------------------------------
# norms
# ---------------------------------------------------------------------------------------------
GN_STATS_WGS = int(_os.environ.get("LGD_GN_STATS_WGS", "1024"))      # workgroups per launch the statistics pass aims at


def gn_chunks(B, HW):
    """Pixel chunks per image of the GroupNorm statistics pass (>= 8 pixels each, at most 64 so that the
    partial table [B, nchunk, G, 2] every apply workgroup re-reduces stays at 16 KB per image)."""
    return max(1, min(HW // 8, GN_STATS_WGS // max(B, 1), 64))


def groupnorm(x, B, HW, G, eps, gamma, beta, silu, *, x1=None, out=None, part=None, stats=None):
    c0 = x.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    C_ = c0 + c1
    nchunk = gn_chunks(B, HW)
    if out is None:
        out = torch.empty((B * HW, C_), device=x.device, dtype=F16)
    if part is None:
        part = torch.empty((B, nchunk, G, 2), device=x.device, dtype=F32)
    # algorithmic bytes: the map read once for the statistics, once for the apply, written once
    _prof("groupnorm (gn_stats + gn_apply | gn_fused)", 6.0 * B * HW * C_,
          lambda: _call("lgd_groupnorm_f16", _p(x), _p(x1), c0, c1, B, HW, G, float(eps), _p(gamma), _p(beta),
                        1 if silu else 0, _p(out), _p(part), nchunk, _p(stats), _stream()), shape=f"B{B}_HW{HW}_C{C_}")
    return out


def groupnorm_bwd(gy, x, B, HW, G, gamma, beta, silu, stats, *, x1=None, gx0=None, gx1=None,
                  part=None, accumulate=False):
    c0 = x.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    nchunk = gn_chunks(B, HW)
    if gx0 is None:
        gx0 = torch.empty_like(x)
    if x1 is not None and gx1 is None:
        gx1 = torch.empty_like(x1)
    if part is None:
        part = torch.empty((B, nchunk, G, 2), device=x.device, dtype=F32)
    _prof("groupnorm_bwd (gn_bwd_stats + gn_bwd_apply)", (10.0 + (2.0 if accumulate else 0.0)) * B * HW * (c0 + c1),
          lambda: _call("lgd_groupnorm_bwd_f16", _p(gy), _p(x), _p(x1), c0, c1, B, HW, G, _p(gamma), _p(beta),
                        1 if silu else 0, _p(stats), _p(gx0), _p(gx1), _p(part), nchunk, 1 if accumulate else 0, _stream()),
          shape=f"B{B}_HW{HW}_C{c0 + c1}")
    return gx0, gx1


def layernorm(x, gamma, beta, eps=1e-5, *, out=None, ldy=None, stats=None, rows_per_batch=0,
              x_bs=0, y_bs=0, rows=None, ldx=None):
    C_ = gamma.shape[0]
    if rows is None:
        rows = x.numel() // C_
    if out is None:
        out = torch.empty((rows, C_), device=x.device, dtype=F16)
    _prof("layernorm_rows_kernel", 4.0 * rows * C_,
          lambda: _call("lgd_layernorm_f16", _p(x), ldx or C_, _p(out), ldy or C_, rows, C_, float(eps), _p(gamma),
                        _p(beta), _p(stats), rows_per_batch, x_bs, y_bs, _stream()), shape=f"R{rows}_C{C_}")
    return out


def layernorm_stats(x, C_, eps=1e-5, *, stats=None, rows=None, ldx=None):
    """(mean, rstd) of every row, fp32 [rows][2]: the statistics half of LayerNorm, whose other half rides in the
    consuming GEMM's epilogue (EPI_ROWNORM)."""
    if rows is None:
        rows = x.numel() // C_
    if stats is None:
        stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    _prof("ln_stats_kernel (LayerNorm statistics only)", 2.0 * rows * C_,
          lambda: _call("lgd_layernorm_f16", _p(x), ldx or C_, _p(None), C_, rows, C_, float(eps), _p(None), _p(None),
                        _p(stats), 0, 0, 0, _stream()), shape=f"R{rows}_C{C_}")
    return stats


def layernorm_bwd(gy, x, gamma, stats, *, gx=None, rows=None, ldgy=None, ldx=None, ldgx=None,
                  rows_per_batch=0, gy_bs=0, x_bs=0, gx_bs=0, accumulate=False):
    C_ = gamma.shape[0]
    if rows is None:
        rows = x.numel() // C_
    if gx is None:
        gx = torch.empty((rows, C_), device=x.device, dtype=F16)
    _prof("layernorm_bwd_kernel", (6.0 + (2.0 if accumulate else 0.0)) * rows * C_,
          lambda: _call("lgd_layernorm_bwd_f16", _p(gy), ldgy or C_, _p(x), ldx or C_, _p(gx), ldgx or C_, rows, C_,
                        _p(gamma), _p(stats), rows_per_batch, gy_bs, x_bs, gx_bs, 1 if accumulate else 0, _stream()),
          shape=f"R{rows}_C{C_}")
    return gx


# ---------------------------------------------------------------------------------------------
# attention.  Views are (tensor, ld, batch_stride) with head h at column offset h*d.
# ---------------------------------------------------------------------------------------------
def attn_fwd(q, k, v, o, B, H, Sq, Sk, d, scale, *, lse=None, q_view=None, k_view=None,
             v_view=None, o_view=None):
    qv = q_view or (H * d, Sq * H * d)
    kv = k_view or (H * d, Sk * H * d)
    vv = v_view or (H * d, Sk * H * d)
    ov = o_view or (H * d, Sq * H * d)
    fn = lambda: _call("lgd_attn_fwd_f16", _p(q), qv[0], qv[1], _p(k), kv[0], kv[1], _p(v), vv[0], vv[1],
                       _p(o), ov[0], ov[1], _p(lse), B, H, Sq, Sk, d, float(scale), _stream())
    if PROFILER is not None:
        PROFILER.wrap(f"attn_self_kernel d={d}", 4.0 * B * H * Sq * Sk * d, 2.0 * B * H * d * (2 * Sq + 2 * Sk), fn,
                      "attn_path", shape=f"B{B}_H{H}_Sq{Sq}_Sk{Sk}")
    else:
        fn()
    return o


def attn_bwd(q, k, v, o, go, lse, delta, gq, gk, gv, B, H, Sq, Sk, d, scale, *, q_view=None,
             k_view=None, v_view=None, o_view=None, go_view=None, gq_view=None, gk_view=None,
             gv_view=None, sk_grad=None):
    """sk_grad: dK / dV are computed (and written) for the first sk_grad keys only (lgd_attn_bwd_keys_f16); default all."""
    skg = Sk if sk_grad is None else int(sk_grad)
    dq_ = (H * d, Sq * H * d)
    dk_ = (H * d, Sk * H * d)
    qv, kv, vv = q_view or dq_, k_view or dk_, v_view or dk_
    ov, gov = o_view or dq_, go_view or dq_
    gqv, gkv, gvv = gq_view or dq_, gk_view or dk_, gv_view or dk_
    # flash backward with recompute: QK^T twice (dQ and dK/dV passes), dP = dO V^T twice, dV, dK, dQ -> 14 B H Sq Sk d
    # executed; the ALGORITHMIC work of an attention backward is 5 contractions = 10 B H Sq Sk d (DESIGN.md kernel table)
    _prof(f"attn_bwd_dq + attn_bwd_dkv d={d}", 2.0 * B * H * d * (4 * Sq + 4 * Sk),
          lambda: _call("lgd_attn_bwd_keys_f16", _p(q), qv[0], qv[1], _p(k), kv[0], kv[1], _p(v), vv[0], vv[1], _p(o),
                        ov[0], ov[1], _p(go), gov[0], gov[1], _p(lse), _p(delta), _p(gq), gqv[0], gqv[1], _p(gk),
                        gkv[0], gkv[1], _p(gv), gvv[0], gvv[1], B, H, Sq, Sk, skg, d, float(scale), _stream()),
          # algorithmic: dQ over all keys (S, dP, dQ) + dK / dV for the keys whose gradient is wanted (2 of the 5 contractions)
          flops=6.0 * B * H * Sq * Sk * d + 4.0 * B * H * Sq * skg * d, tag="attn_path_bwd", shape=f"B{B}_H{H}_Sq{Sq}_Sk{Sk}")


def cross_attn_fwd(q, k, v, o, B, H, Sq, Sk, d, scale, *, probs=None, tok=-1, cond_only=False,
                   q_view=None, k_view=None, v_view=None, o_view=None):
    qv = q_view or (H * d, Sq * H * d)
    kv = k_view or (H * d, Sk * H * d)
    vv = v_view or (H * d, Sk * H * d)
    ov = o_view or (H * d, Sq * H * d)
    fn = lambda: _call("lgd_cross_attn_fwd_f16", _p(q), qv[0], qv[1], _p(k), kv[0], kv[1], _p(v), vv[0], vv[1],
                       _p(o), ov[0], ov[1], _p(probs), int(tok), 1 if cond_only else 0, B, H, Sq, Sk, d,
                       float(scale), _stream())
    if PROFILER is not None:
        PROFILER.wrap(f"attn_fwd_kernel(cross) d={d}", 4.0 * B * H * Sq * Sk * d,
                      2.0 * B * H * d * (2 * Sq + 2 * Sk), fn, "attn_path")
    else:
        fn()
    return o


def attn_causal_fwd(q, k, v, o, B, H, S, d, scale, *, view=None):
    """Causal self-attention over a fused [B,S,3*H*d] projection (or separate q/k/v with the default view)."""
    vw = view or (H * d, S * H * d)
    _call("lgd_attn_causal_fwd_f16", _p(q), vw[0], vw[1], _p(k), vw[0], vw[1], _p(v), vw[0], vw[1], _p(o),
          H * d, S * H * d, B, H, S, d, float(scale), _stream())
    return o


def quick_gelu(x, out=None):
    if out is None:
        out = torch.empty_like(x)
    _call("lgd_quick_gelu_f16", _p(x), _p(out), x.numel(), _stream())
    return out


ACT_GELU, ACT_RELU = 1, 2


def act(x, mode, out=None):
    """Exact GELU / ReLU on an fp16 tensor (SAM encoder and mask-decoder MLPs)."""
    if out is None:
        out = torch.empty_like(x)
    _call("lgd_act_f16", _p(x), _p(out), x.numel(), int(mode), _stream())
    return out


def sam_relpos_qkv(qkv, qkv_bias, rel_h, rel_w, B, Hs, Ws, window, NH, d, DA, scale):
    """Window partition + decomposed rel-pos bias folded into the attention operands (csrc/sam.hip).
    Returns qa, ka, va [B*nwin*S*S, NH*DA] fp16."""
    S = window or Hs
    rows = B * (-(-Hs // S)) * (-(-Ws // S)) * S * S
    qa, ka, va = (torch.empty((rows, NH * DA), device=qkv.device, dtype=F16) for _ in range(3))
    _call("lgd_sam_relpos_qkv_f16", _p(qkv), _p(qkv_bias), _p(rel_h), _p(rel_w), B, Hs, Ws, window, NH, d, DA,
          float(scale), _p(qa), _p(ka), _p(va), _stream())
    return qa, ka, va


def sam_window_merge(oa, B, Hs, Ws, window, NH, d, DA):
    out = torch.empty((B * Hs * Ws, NH * d), device=oa.device, dtype=F16)
    _call("lgd_sam_window_merge_f16", _p(oa), _p(out), B, Hs, Ws, window, NH, d, DA, _stream())
    return out


def cross_attn_bwd(q, k, v, go, gp, gq, B, H, Sq, Sk, d, scale, *, q_view=None, k_view=None,
                   v_view=None, go_view=None, gq_view=None):
    dq_ = (H * d, Sq * H * d)
    dk_ = (H * d, Sk * H * d)
    qv, kv, vv = q_view or dq_, k_view or dk_, v_view or dk_
    gov, gqv = go_view or dq_, gq_view or dq_
    # dQ only (text K / V are constants of the run): recomputed scores, dP = dO V^T (+ the map gradient), dQ = dS K
    _prof(f"cross_attn_bwd_mfma_kernel d={d}", 2.0 * B * H * d * (3 * Sq + 2 * Sk) + (4.0 * B * H * Sq * Sk if gp is not None else 0.0),
          lambda: _call("lgd_cross_attn_bwd_f16", _p(q), qv[0], qv[1], _p(k), kv[0], kv[1], _p(v), vv[0], vv[1],
                        _p(go), gov[0], gov[1], _p(gp), _p(gq), gqv[0], gqv[1], B, H, Sq, Sk, d, float(scale), _stream()),
          flops=(6.0 if go is not None else 4.0) * B * H * Sq * Sk * d, tag="attn_path_bwd")
    return gq


# ---------------------------------------------------------------------------------------------
# elementwise / step
# ---------------------------------------------------------------------------------------------
def geglu_bwd(h, gy, gh=None):
    rows, n2 = h.shape
    if gh is None:
        gh = torch.empty_like(h)
    _prof("geglu_bwd_kernel", 2.0 * rows * (n2 + n2 // 2 + n2),
          lambda: _call("lgd_geglu_bwd_f16", _p(h), _p(gy), _p(gh), rows, n2 // 2, _stream()))
    return gh


def geglu_fwd(h, out=None):
    rows, n2 = h.shape
    if out is None:
        out = torch.empty((rows, n2 // 2), device=h.device, dtype=F16)
    _prof("geglu_fwd_kernel", 2.0 * rows * (n2 + n2 // 2),
          lambda: _call("lgd_geglu_fwd_f16", _p(h), _p(out), rows, n2 // 2, _stream()))
    return out


def softmax_rows(x, scale=1.0, out=None):
    rows, n = x.shape
    if out is None:
        out = torch.empty_like(x)
    _call("lgd_softmax_rows_f16", _p(x), _p(out), rows, n, float(scale), _stream())
    return out


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    _prof("add_kernel", 6.0 * a.numel(), lambda: _call("lgd_add_f16", _p(a), _p(b), _p(out), a.numel(), _stream()))
    return out


def scale(x, alpha, out=None):
    if out is None:
        out = torch.empty_like(x)
    _call("lgd_scale_f16", _p(x), _p(out), float(alpha), x.numel(), _stream())
    return out


def upsample2x_bwd(gy, B, H, W, C_, out=None):
    if out is None:
        out = torch.empty((B * H * W, C_), device=gy.device, dtype=F16)
    _prof("upsample2x_bwd_kernel", 2.0 * 5 * B * H * W * C_,
          lambda: _call("lgd_upsample2x_bwd_f16", _p(gy), _p(out), B, H, W, C_, _stream()))
    return out


def cfg_ddim_step(eps, x, x_out, coef_table, dyn, *, frozen_ref=None, mask=None, hist=None):
    """dyn: device int32[2] = {step, frozen_steps}."""
    B, C_, L, _ = x.shape
    _prof("cfg_ddim_kernel", 4.0 * x.numel() * 6,
          lambda: _call("lgd_cfg_ddim_step_f32", _p(eps), _p(x), _p(x_out), _p(coef_table), _p(dyn),
                        _p(frozen_ref), _p(mask), _p(hist), B, C_, L * L, _stream()))
    return x_out


def cfg_multistep_step(eps, x, x_out, x0_prev, coef_table, dyn, *, frozen_ref=None, mask=None, hist=None):
    """CFG + one linear multistep update (DPM-Solver++ 2M); coef_table fp32 [T][8], see lgd_hip.h."""
    B, C_, L, _ = x.shape
    _call("lgd_cfg_multistep_step_f32", _p(eps), _p(x), _p(x_out), _p(x0_prev), _p(coef_table), _p(dyn),
          _p(frozen_ref), _p(mask), _p(hist), B, C_, L * L, _stream())
    return x_out


def axpy(g, x, coef_table, step_idx, col, active=None):
    per = x.numel() // x.shape[0] if active is not None else 0
    _prof("axpy_kernel", 12.0 * x.numel(),
          lambda: _call("lgd_axpy_f32", _p(g), _p(x), _p(coef_table), _p(step_idx), int(col), _p(active), per, x.numel(),
                        _stream()))


def scale_rows(x, out, table, dyn, col, reps=1):
    """out[r] = x * table[dyn[0]][col], r < reps (EulerDiscrete.scale_model_input for the CFG pair; lgd_hip.h)."""
    _call("lgd_scale_rows_f32", _p(x), _p(out), _p(table), _p(dyn), int(table.shape[1]), int(col), x.numel(), int(reps),
          _stream())
    return out


def select_row(table, idx, out):
    _prof("select_row_kernel", 8.0 * out.numel(),
          lambda: _call("lgd_select_row_f32", _p(table), _p(idx), _p(out), out.numel(), _stream()))


def ca_energy(map_ptrs, gmap_ptrs, map_hw, items, coefs, masks, refs, refs_step_stride, dyn, groups, n_groups,
              n_items, H, T, max_hw, partial, loss, grad_scale=1.0, n_samples=1):
    _prof("ca_energy_kernel + energy_sum_kernel", 4.0 * n_groups * H * max_hw * 2,
          lambda: _call("lgd_ca_energy_f32", _p(map_ptrs), _p(gmap_ptrs), _p(map_hw), _p(items), _p(coefs),
                        _p(masks), _p(refs), int(refs_step_stride), _p(dyn), _p(groups), n_groups, n_items, n_samples, H, T,
                        max_hw, float(grad_scale), _p(partial), _p(loss), _stream()))


def boxdiff_energy(map_ptrs, gmap_ptrs, n_maps, side, items, masks, smooth, groups, n_samples, max_items, H, T,
                   loss_scale, grad_scale, loss):
    """BoxDiff energy + map gradients (lgd_boxdiff_energy_f32, include/lgd_hip.h; utils/boxdiff.py:20-196)."""
    _prof("boxdiff_energy_kernel", 4.0 * 2 * n_maps * n_samples * H * side * side * T * 2,
          lambda: _call("lgd_boxdiff_energy_f32", _p(map_ptrs), _p(gmap_ptrs), int(n_maps), int(side), _p(items), _p(masks),
                        _p(smooth), _p(groups), int(n_samples), int(max_items), int(H), int(T), float(loss_scale),
                        float(grad_scale), _p(loss), _stream()))
