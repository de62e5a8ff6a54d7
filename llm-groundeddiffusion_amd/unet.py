"""HIP UNet engine: the SD 1.x / 2.x (+GLIGEN) UNet2DConditionModel forward — and its input-gradient
for backward guidance — as a static plan of C-ABI kernel launches over channels-last fp16 buffers.

Replaces models/unet_2d_condition.py:704-980 + unet_2d_blocks.py + transformer_2d.py + attention.py +
attention_processor.py of the reference (and torch.autograd for pipelines.py:56).  Design:

  * a Plan is built once per (batch, grad?, fuser on/off): every activation gets its own
    pre-allocated buffer, every op is a closure over raw device pointers; running a plan is a
    flat loop of kernel launches (hipGraph-capturable: no allocation, no sync, no host reads);
  * layout is channels-last [B*HW, C] everywhere, so the NCHW<->(B,HW,C) permutes of
    transformer_2d.py:287,321 disappear and 1x1 convs are plain GEMMs; torch.cat of skip
    connections (unet_2d_blocks.py:646-649) is never materialised (two-source GEMM/GroupNorm);
  * per-run constants are hoisted out of the 50-step loop: text K/V of all 16 cross-attention
    layers, the time-embedding projections of all resnets for all timesteps, GLIGEN grounding
    tokens (position_net + fuser.linear + LayerNorm) — `prepare_run`;
  * the backward plan is explicit (no autograd): dgrad GEMMs/convs with pre-transposed weights,
    flash-attention backward, GroupNorm/LayerNorm/GEGLU backward kernels, gradient fan-in resolved
    at plan-build time (first writer overwrites, later writers accumulate); it stops where the
    reference's loss stops depending on activations: the cross-attention map of the last guidance
    key (pipelines.py:46 TODO).
"""
import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from ._lib import EPI_GEGLU
from .weights import UNetConfig, unet_blocks
from .weightstore import WeightStore

F16, F32 = torch.float16, torch.float32
N_OBJ_TOKENS = 30  # max_objs of pipelines.py:289


class Act:
    """An activation buffer [rows, C] (fp16) and, in grad plans, its gradient buffer."""
    __slots__ = ("t", "g", "rows", "C")

    def __init__(self, t):
        self.t = t
        self.g = None
        self.rows, self.C = t.shape


class Op:
    def __init__(self, fwd, make_bwd=None, gouts=(), passthrough=None):
        self.fwd = fwd
        self.make_bwd = make_bwd    # (acc flags for gouts) -> callable
        self.gouts = list(gouts)    # Acts whose .g this op's backward writes, in write order
        self.acc = None
        self.bwd = None
        # (res, y): this op computes y = f(x) + res, so d/d res = d/d y unchanged.  When this op is the FIRST writer of
        # res.g, res.g simply IS y.g's buffer (Plan._finalize_backward) instead of a copy of it: y.g is complete when this
        # op's backward runs and is never read after it, so later accumulations into it are safe
        self.passthrough = passthrough


choose_splits = ops.choose_splits


class Plan:
    def __init__(self, eng: "UNetEngine", B: int, L: int, *, grad: bool, fuser: bool,
                 stop_key: Optional[Tuple] = None, save_keys: Sequence[Tuple] = (),
                 text_batch_offset: int = 0, obj_batch_offset: int = 0):
        self.eng, self.B, self.L = eng, B, L
        self.obj_off = obj_batch_offset
        self.grad, self.fuser = grad, fuser
        self.stop_key = tuple(stop_key) if stop_key else None
        self.save_keys = [tuple(k) for k in save_keys]
        self.text_off = text_batch_offset
        self.ops: List[Op] = []
        self._buffers: List[torch.Tensor] = []
        self.dbg: Dict[str, Act] = {}                 # named activations (debugging / tests)
        self.maps: Dict[Tuple, torch.Tensor] = {}     # key -> fp32 [B,H,HW,T] captured probabilities
        self.gmaps: Dict[Tuple, torch.Tensor] = {}    # key -> fp32 gradient of the map (grad plans)
        self._cursor = [0, 0]                         # (segment, byte offset) of the shared activation arena
        self.arena_bytes = 0
        self.latents_in = self._alloc((B, eng.cfg.in_channels, L, L))
        self.eps_out = None
        self.g_latents = None
        self._build()

    # --------------------------------------------------------------------------------------
    def _new(self, rows, C, dtype=F16):
        return self._alloc((rows, C), dtype)

    def _alloc(self, shape, dtype=F32):
        """Every buffer of a plan is carved out of the engine's activation arena, starting at offset 0 of
        segment 0 for EVERY plan: plans alias each other.  Only one plan is in flight at a time and nothing
        a plan produces is read after another plan ran (maps / noise prediction / latent gradient are
        consumed by the caller right after the launch sequence; grad plans keep their forward activations
        until their own backward), so the resident footprint is the largest plan, not the sum of all
        (batch, grad, fuser) variants — and addresses stay fixed, as captured hipGraphs need."""
        t, self._cursor = self.eng.arena_take(self._cursor, shape, dtype)
        self.arena_bytes += t.numel() * t.element_size()
        self._buffers.append(t)
        return t

    def _act(self, rows, C):
        return Act(self._new(rows, C))

    def _add(self, fwd, make_bwd=None, gouts=(), passthrough=None):
        self.ops.append(Op(fwd, make_bwd if self.grad else None, gouts if self.grad else (),
                           passthrough if self.grad else None))

    def _ws(self, n_floats):
        return self.eng.workspace(n_floats)

    # ---- GEMM-shaped ops ------------------------------------------------------------------
    def linear(self, x: Act, name: str, *, res: Optional[Act] = None, alpha=1.0, geglu=False,
               bias=True, out: Optional[Act] = None, rows=None, bwd=True) -> Act:
        """y = alpha*(x @ W^T + b) + res ;  backward: gx (+)= alpha * gy @ W ; gres (+)= gy."""
        W = self.eng.w.h[f"{name}.w"]
        b = self.eng.w.f[f"{name}.b"] if bias else None
        N, K = W.shape
        M = rows or x.rows
        n_out = N // 2 if geglu else N
        y = out or self._act(M, n_out)
        d = ops.gemm_desc(x.t, W, y.t, M, N, K, lda0=x.C, bias=b, res=res.t if res else None,
                          ldr=res.C if res else 0, alpha=alpha, epi=EPI_GEGLU if geglu else 0,
                          ldc=y.C)
        # q/k/v/out projections count towards the "attention path" of the benchmark's roofline report
        tag = "attn_path" if any(t in name for t in (".attn1.", ".attn2.", ".fuser.attn.")) else None
        fwd = lambda: ops.gemm_launch(d, tag)
        if not (self.grad and bwd):
            self._add(fwd)
            return y
        assert not geglu, "grad plans keep the GEGLU pre-activation (see ff())"
        Wt = self.eng.w.h[f"{name}.wt"]
        gouts = [x] + ([res] if res else [])

        def make_bwd(acc):
            dd = ops.gemm_desc(y.g, Wt, x.g, M, K, N, lda0=y.C, alpha=alpha, ldc=x.C,
                               res=x.g if acc[0] else None, ldr=x.C)
            if res is None:
                return lambda: ops.gemm_launch(dd)
            racc = acc[1]
            aliased = res.g.data_ptr() == y.g.data_ptr()

            def run():
                ops.gemm_launch(dd)
                if racc:
                    ops.add(res.g, y.g, res.g)
                elif not aliased:
                    ops.copy_(res.g, y.g)
            return run
        self._add(fwd, make_bwd, gouts, passthrough=(res, y) if res is not None else None)
        return y

    def conv(self, x: Act, name: str, H: int, *, x1: Optional[Act] = None, res: Optional[Act] = None,
             temb_off: Optional[int] = None, stride=1, ups=False) -> Act:
        """3x3 conv (pad 1) on a channels-last map of side H (stored), optional second source."""
        B = self.B
        W = self.eng.w.h[f"{name}.w"]
        b = self.eng.w.f[f"{name}.b"]
        Cout, K = W.shape
        c0, c1 = x.C, (x1.C if x1 else 0)
        Hl = 2 * H if ups else H
        Ho = (Hl - 1) // stride + 1
        M = B * Ho * Ho
        y = self._act(M, Cout)
        eng = self.eng
        if temb_off is not None and eng.temb_rows > 1:
            # text_time conditioning (SDXL): the time embedding differs per image (pooled text + size / score ids), the
            # GEMM epilogue takes ONE bias2 vector per launch -> one launch per image (1024^2 refiner: 16384 rows each)
            assert stride == 1 and not ups and res is None
            ds = []
            for i in range(B):
                row = eng.temb_cur[self.text_off + i]
                sl = slice(i * H * H, (i + 1) * H * H)
                ds.append(ops.gemm_desc(x.t[sl], W, y.t[sl], H * H, Cout, K, a1=x1.t[sl] if x1 else None, c0=c0, c1=c1,
                                        lda0=c0, lda1=c1, taps=9, hin=H, win=H, hout=H, wout=H, bias=b,
                                        bias2=row[temb_off:temb_off + Cout], ldc=Cout))
            fwd = lambda: [ops.gemm_launch(dd) for dd in ds]
        else:
            bias2 = eng.temb_cur[0, temb_off:temb_off + Cout] if temb_off is not None else None
            d = ops.gemm_desc(x.t, W, y.t, M, Cout, K, a1=x1.t if x1 else None, c0=c0, c1=c1, lda0=c0,
                              lda1=c1, taps=9, hin=H, win=H, hout=Ho, wout=Ho, stride=stride,
                              ups=1 if ups else 0, bias=b, bias2=bias2, res=res.t if res else None,
                              ldr=res.C if res else 0, ldc=Cout)
            fwd = lambda: ops.gemm_launch(d)
        if not self.grad:
            self._add(fwd)
            return y
        Wd = self.eng.w.h[f"{name}.wd"]          # [Cin][9*Cout]
        gouts = [x] + ([x1] if x1 else []) + ([res] if res else [])

        def make_bwd(acc):
            runs = []
            Kd = 9 * Cout
            if stride == 1 and not ups:
                # dIn = conv(dOut, flipped W^T); one launch per source (channel slice of Wd rows)
                for i, (src, lo, cn) in enumerate([(x, 0, c0)] + ([(x1, c0, c1)] if x1 else [])):
                    dd = ops.gemm_desc(y.g, Wd[lo:lo + cn], src.g, src.rows, cn, Kd, c0=Cout, lda0=Cout,
                                       taps=9, hin=H, win=H, hout=H, wout=H, ldc=cn,
                                       res=src.g if acc[i] else None, ldr=cn)
                    runs.append(lambda dd=dd: ops.gemm_launch(dd))
            elif stride == 2:
                # zero-inserted gather of dOut (hin = Ho) produces the H x H input gradient
                dd = ops.gemm_desc(y.g, Wd, x.g, x.rows, c0, Kd, c0=Cout, lda0=Cout, taps=9, hin=Ho,
                                   win=Ho, hout=H, wout=H, ups=2, ldc=c0,
                                   res=x.g if acc[0] else None, ldr=c0)
                runs.append(lambda: ops.gemm_launch(dd))
            else:
                # nearest-2x upsample folded in forward: dgrad at 2H x 2H, then 2x2 sum
                tmp = self._new(B * Hl * Hl, c0)
                dd = ops.gemm_desc(y.g, Wd, tmp, B * Hl * Hl, c0, Kd, c0=Cout, lda0=Cout, taps=9,
                                   hin=Hl, win=Hl, hout=Hl, wout=Hl, ldc=c0)
                assert not acc[0]
                runs.append(lambda: (ops.gemm_launch(dd), ops.upsample2x_bwd(tmp, B, H, H, c0, out=x.g)))
            if res is not None:
                racc = acc[-1]
                if racc:
                    runs.append(lambda: ops.add(res.g, y.g, res.g))
                elif res.g.data_ptr() != y.g.data_ptr():
                    runs.append(lambda: ops.copy_(res.g, y.g))
            return lambda: [r() for r in runs]
        self._add(fwd, make_bwd, gouts, passthrough=(res, y) if res is not None else None)
        return y

    def shortcut(self, x: Act, x1: Optional[Act], name: str) -> Act:
        """1x1 conv_shortcut on the (virtually concatenated) resnet input."""
        W = self.eng.w.h[f"{name}.w"]
        b = self.eng.w.f[f"{name}.b"]
        Cout, K = W.shape
        c0, c1 = x.C, (x1.C if x1 else 0)
        M = x.rows
        y = self._act(M, Cout)
        d = ops.gemm_desc(x.t, W, y.t, M, Cout, K, a1=x1.t if x1 else None, c0=c0, c1=c1, lda0=c0,
                          lda1=c1, bias=b, ldc=Cout)
        fwd = lambda: ops.gemm_launch(d)
        if not self.grad:
            self._add(fwd)
            return y
        Wt = self.eng.w.h[f"{name}.wt"]          # [Cin][Cout]
        gouts = [x] + ([x1] if x1 else [])

        def make_bwd(acc):
            runs = []
            for i, (src, lo, cn) in enumerate([(x, 0, c0)] + ([(x1, c0, c1)] if x1 else [])):
                dd = ops.gemm_desc(y.g, Wt[lo:lo + cn], src.g, M, cn, Cout, lda0=Cout, ldc=cn,
                                   res=src.g if acc[i] else None, ldr=cn)
                runs.append(lambda dd=dd: ops.gemm_launch(dd))
            return lambda: [r() for r in runs]
        self._add(fwd, make_bwd, gouts)
        return y

    # ---- norms -----------------------------------------------------------------------------
    def groupnorm(self, x: Act, x1: Optional[Act], name: str, HW: int, eps: float, silu: bool) -> Act:
        eng, B = self.eng, self.B
        G = eng.cfg.norm_num_groups
        gm, bt = eng.w.f[f"{name}.g"], eng.w.f[f"{name}.b"]
        C = x.C + (x1.C if x1 else 0)
        y = self._act(x.rows, C)
        nch = ops.gn_chunks(B, HW)
        part = self._alloc((B, nch, G, 2))
        stats = self._alloc((B, G, 2)) if self.grad else None
        xt1 = x1.t if x1 else None
        fwd = lambda: ops.groupnorm(x.t, B, HW, G, eps, gm, bt, silu, x1=xt1, out=y.t, part=part, stats=stats)
        if not self.grad:
            self._add(fwd)
            return y
        gouts = [x] + ([x1] if x1 else [])

        def make_bwd(acc):
            a = acc[0]
            if x1 is not None:
                assert acc[0] == acc[1], "mixed accumulate modes on a concat GroupNorm"
            return lambda: ops.groupnorm_bwd(y.g, x.t, B, HW, G, gm, bt, silu, stats, x1=xt1, gx0=x.g,
                                             gx1=x1.g if x1 else None, part=part, accumulate=a)
        self._add(fwd, make_bwd, gouts)
        return y

    def layernorm(self, x: Act, name: str, *, out_t=None, ldy=None, S=None, y_bs=0) -> Act:
        """LayerNorm rows of x.  With out_t/ldy/S/y_bs the rows of each image are written at the
        head of a larger [S+30] buffer (GLIGEN fuser concat)."""
        eng = self.eng
        gm, bt = eng.w.f[f"{name}.g"], eng.w.f[f"{name}.b"]
        C = x.C
        y = Act(out_t) if out_t is not None else self._act(x.rows, C)
        stats = self._alloc((x.rows, 2)) if self.grad else None
        rpb = S or 0
        x_bs = (S or 0) * C
        fwd = lambda: ops.layernorm(x.t, gm, bt, out=y.t, ldy=ldy or C, stats=stats, rows=x.rows,
                                    rows_per_batch=rpb, x_bs=x_bs, y_bs=y_bs)
        if not self.grad:
            self._add(fwd)
            return y

        def make_bwd(acc):
            return lambda: ops.layernorm_bwd(y.g, x.t, gm, stats, gx=x.g, rows=x.rows, ldgy=ldy or C,
                                             rows_per_batch=rpb, gy_bs=y_bs, x_bs=x_bs, gx_bs=x_bs,
                                             accumulate=acc[0])
        self._add(fwd, make_bwd, [x])
        return y

    def ln_linear(self, x: Act, norm: str, name: str, *, geglu=False) -> Act:
        """linear(LayerNorm(x)) (attention.py:185,206,223 + the projection behind each).  No-grad plans of an engine with
        `fold_ln` read the rows RAW: one statistics pass (mean, rstd per row; no normalised copy is written or re-read)
        and the normalisation rides in the GEMM's epilogue on the gamma-folded weights (weightstore._lnlin,
        LGD_EPI_ROWNORM).  Grad plans keep the two ops: the backward needs the LayerNorm's own node."""
        eng = self.eng
        has_b = f"{name}.b" in eng.w.f
        # the statistics-only form of lgd_layernorm_f16 (y = NULL) exists in the row kernels only: widths up to 1536
        # (every SD 1.x / 2.x / SDXL-refiner transformer); wider rows keep the two ops
        if self.grad or not eng.fold_ln or x.C > 1536:
            y = self.layernorm(x, norm)
            if geglu:
                return self.linear(y, name, geglu=True)
            return self.linear(y, name, bias=has_b)
        W, cs, b = eng.w.h[f"{name}.wln"], eng.w.f[f"{name}.cs"], eng.w.f[f"{name}.bln"]
        N, K = W.shape
        M = x.rows
        stats = self._alloc((M, 2))
        y = self._act(M, N // 2 if geglu else N)
        d = ops.gemm_desc(x.t, W, y.t, M, N, K, lda0=x.C, bias=b, epi=EPI_GEGLU if geglu else 0, ldc=y.C,
                          rowstat=stats, colsum=cs)
        tag = "attn_path" if any(t in name for t in (".attn1.", ".attn2.", ".fuser.attn.")) else None
        xt = x.t
        self._add(lambda: (ops.layernorm_stats(xt, K, stats=stats, rows=M), ops.gemm_launch(d, tag)))
        return y

    # ---- attention -------------------------------------------------------------------------
    def self_attn(self, qkv: Act, heads: int, S: int, Sk: Optional[int] = None) -> Act:
        """Flash attention over a fused [B*(Sk), 3C] projection; queries are the first S rows of each
        image (Sk = S+30 for the GLIGEN fuser, attention.py:50)."""
        B = self.B
        Sk = Sk or S
        C = qkv.C // 3
        d = C // heads
        o = self._act(B * S, C)
        view = (3 * C, Sk * 3 * C)
        lse = self._alloc((B, heads, S)) if self.grad else None
        scale = d ** -0.5
        qt, kt, vt = qkv.t, qkv.t[:, C:], qkv.t[:, 2 * C:]
        fwd = lambda: ops.attn_fwd(qt, kt, vt, o.t, B, heads, S, Sk, d, scale, lse=lse,
                                   q_view=view, k_view=view, v_view=view)
        if not self.grad:
            self._add(fwd)
            return o

        def make_bwd(acc):
            assert not acc[0]
            delta = self._alloc((B, heads, S))
            g = qkv.g
            gq, gk, gv = g, g[:, C:], g[:, 2 * C:]
            pad = Sk > S

            # the gradient rows of the 30 grounding tokens are cleared whole (round 6): they have no queries, and their
            # key / value gradients are not computed — those rows of the concatenated input are constants of a run
            # (nothing reads the gradient behind them: the LayerNorm backward takes the visual rows only), and the key
            # block holding them cost the dK/dV pass an extra round of workgroups (sk_grad = S: lgd_attn_bwd_keys_f16).
            # The backward kernels overwrite every other element of the fused [q | k | v] gradient
            g_tail = g.view(B, Sk, 3 * C)[:, S:, :] if pad else None

            def run():
                if pad:
                    ops.zero_(g_tail)
                ops.attn_bwd(qt, kt, vt, o.t, o.g, lse, delta, gq, gk, gv, B, heads, S, Sk, d, scale,
                             q_view=view, k_view=view, v_view=view, gq_view=view, gk_view=view, gv_view=view, sk_grad=S)
            return run
        self._add(fwd, make_bwd, [qkv])
        return o

    def cross_attn(self, q: Act, key: Tuple, layer_name: str, heads: int, S: int, last: bool) -> Optional[Act]:
        eng, B = self.eng, self.B
        C = q.C
        d = C // heads
        T = eng.text_len
        kv = eng.text_kv[layer_name]                    # [Bt, 77, 2C]; layer_name = UNetEngine.kv_name(prefix, depth index)
        kv_t = kv[self.text_off:]
        k_view = (2 * C, T * 2 * C)
        o = self._act(B * S, C)
        scale = d ** -0.5
        probs = None
        if key in self.save_keys:
            # the map is always captured whole (all images, all 77 columns) into a static buffer, so
            # the launch has no per-step arguments; callers slice what attention_processor.py:466-476
            # would have kept (token column / conditional half) when they copy it out.
            probs = self.maps[key] = self._alloc((B, heads, S, T))
            if self.grad:
                self.gmaps[key] = self._alloc((B, heads, S, T))   # zeroed by the energy launch (EnergyTables.run)
        kt, vt = kv_t, kv_t[:, :, C:]

        def fwd():
            ops.cross_attn_fwd(q.t, kt, vt, o.t, B, heads, S, T, d, scale, probs=probs,
                               k_view=k_view, v_view=k_view)
        if not self.grad:
            self._add(fwd)
            return o
        gp = self.gmaps.get(key)

        def make_bwd(acc):
            assert not acc[0]
            return lambda: ops.cross_attn_bwd(q.t, kt, vt, None if last else o.g, gp, q.g, B, heads, S, T,
                                              d, scale, k_view=k_view, v_view=k_view)
        self._add(fwd, make_bwd, [q])
        return o

    def ff(self, x: Act, res: Act, name: str, alpha=1.0, norm: Optional[str] = None) -> Act:
        """FeedForward(GEGLU) + residual (attention.py:228-233): fused epilogue in no-grad plans.  norm: the LayerNorm
        in front of it (x is then its INPUT)."""
        if not self.grad:
            h = self.ln_linear(x, norm, f"{name}.net.0.proj", geglu=True) if norm else self.linear(x, f"{name}.net.0.proj", geglu=True)
            return self.linear(h, f"{name}.net.2", res=res, alpha=alpha)
        if norm:
            x = self.layernorm(x, norm)
        pre = self.linear(x, f"{name}.net.0.proj")                 # packed [M, 8C] pre-activation
        act = self._act(x.rows, pre.C // 2)
        self._add(lambda: ops.geglu_fwd(pre.t, out=act.t),
                  lambda acc: (lambda: ops.geglu_bwd(pre.t, act.g, pre.g)), [pre])
        return self.linear(act, f"{name}.net.2", res=res, alpha=alpha)

    # --------------------------------------------------------------------------------------
    def _resnet(self, r, x: Act, skip: Optional[Act], H: int) -> Act:
        eng = self.eng
        eps = eng.cfg.norm_eps
        HW = H * H
        h = self.groupnorm(x, skip, f"{r.prefix}.norm1", HW, eps, True)
        h = self.conv(h, f"{r.prefix}.conv1", H, temb_off=eng.w.temb_offsets[r.prefix])
        h = self.groupnorm(h, None, f"{r.prefix}.norm2", HW, eps, True)
        if r.shortcut:
            sc = self.shortcut(x, skip, f"{r.prefix}.conv_shortcut")
        else:
            assert skip is None
            sc = x
        out = self.conv(h, f"{r.prefix}.conv2", H, res=sc)
        self.dbg[f"{r.prefix}.out"] = out
        return out

    def _transformer(self, a, x: Act, H: int) -> Optional[Act]:
        eng, B = self.eng, self.B
        S = H * H
        C, heads = a.channels, a.heads
        n = self.groupnorm(x, None, f"{a.prefix}.norm", S, 1e-6, False)          # transformer_2d.py:283
        h = self.linear(n, f"{a.prefix}.proj_in")
        for dpt in range(a.depth):                                               # transformer_2d.py:293-302
            t = f"{a.prefix}.transformer_blocks.{dpt}"
            key = tuple(a.key[:3]) + (dpt,)
            # 1. self-attention (attention.py:185-195)
            qkv = self.ln_linear(h, f"{t}.norm1", f"{t}.attn1.qkv")
            h = self.linear(self.self_attn(qkv, heads, S), f"{t}.attn1.to_out.0", res=h)
            if dpt == 0:
                self.dbg[f"{a.prefix}.after_attn1"] = h
            # 1.5 GLIGEN gated self-attention (attention.py:43-53, 198-200)
            if self.fuser:
                f = f"{t}.fuser"
                Sk = S + N_OBJ_TOKENS
                cat = eng.fuser_cat(eng.kv_name(a.prefix, dpt), B, S, self.obj_off)   # [B*(S+30), C], tail rows preset
                cat_act = self.layernorm(h, f"{f}.norm1", out_t=cat, ldy=C, S=S, y_bs=Sk * C)
                # the gradient w.r.t. the concat buffer is consumed only by the LayerNorm backward of the
                # visual rows; the 30 grounding rows are constants of the run
                qkv_f = self.linear(cat_act, f"{f}.attn.qkv", bias=False)
                o = self.self_attn(qkv_f, heads, S, Sk)
                h = self.linear(o, f"{f}.attn.to_out.0", res=h, alpha=eng.w.scalars[f"{f}.alpha_attn"])
                self.dbg[f"{a.prefix}.after_fuser_attn"] = h
                h = self.ff(h, h, f"{f}.ff", alpha=eng.w.scalars[f"{f}.alpha_dense"], norm=f"{f}.norm2")
                self.dbg[f"{a.prefix}.after_fuser"] = h
            # 2. cross-attention (attention.py:204-220) — the hook of attention_processor.py:377-483
            q = self.ln_linear(h, f"{t}.norm2", f"{t}.attn2.to_q")
            last = self.stop_key is not None and key == self.stop_key
            o = self.cross_attn(q, key, eng.kv_name(a.prefix, dpt), heads, S, last)
            if last:
                return None
            h = self.linear(o, f"{t}.attn2.to_out.0", res=h)
            if dpt == 0:
                self.dbg[f"{a.prefix}.after_attn2"] = h
            # 3. feed-forward (attention.py:223-233)
            h = self.ff(h, h, f"{t}.ff", norm=f"{t}.norm3")
        out = self.linear(h, f"{a.prefix}.proj_out", res=x)                      # transformer_2d.py:319-327
        self.dbg[f"{a.prefix}.out"] = out
        return out

    def _build(self):
        eng, B, L = self.eng, self.B, self.L
        cfg = eng.cfg
        w = eng.w
        blocks = eng.blocks
        c0 = cfg.block_out_channels[0]
        x = self._act(B * L * L, c0)
        lat = self.latents_in
        self._x0 = x
        x0 = x
        # conv_in on the matrix cores: latents -> 8-channel fp16 map (4 zero channels), 3x3 implicit GEMM with K = 72
        lat8 = self._new(B * L * L, 8)
        d_in = ops.gemm_desc(lat8, w.h["conv_in.w8"], x0.t, B * L * L, c0, 72, c0=8, lda0=8, taps=9, hin=L, win=L, hout=L,
                             wout=L, bias=w.f["conv_in.b"], ldc=c0, splits=1)
        f_in = 2.0 * B * L * L * c0 * 9 * cfg.in_channels                   # algorithmic: the filter has 9 x 4 taps, not 72
        self._add(lambda: (ops.nchw_to_nhwc8(lat, out=lat8), ops.gemm_launch(d_in, None, f_in)))
        skips = [(x, L)]
        H = L
        done = False
        for b in blocks:
            if done:
                break
            if b.kind == "down":
                for j, r in enumerate(b.resnets):
                    x = self._resnet(r, x, None, H)
                    if b.attns:
                        x = self._transformer(b.attns[j], x, H)
                        if x is None:            # the last saved / guidance key is a down-level one: the plan ends here
                            done = True
                            break
                    skips.append((x, H))
                if done:
                    break
                if b.sampler:
                    x = self.conv(x, b.sampler, H, stride=2)
                    H = (H - 1) // 2 + 1
                    skips.append((x, H))
            elif b.kind == "mid":
                x = self._resnet(b.resnets[0], x, None, H)
                x = self._transformer(b.attns[0], x, H)
                if x is None:
                    done = True
                    break
                x = self._resnet(b.resnets[1], x, None, H)
            else:
                for j, r in enumerate(b.resnets):
                    sk, _ = skips.pop()
                    x = self._resnet(r, x, sk, H)
                    if b.attns:
                        x = self._transformer(b.attns[j], x, H)
                        if x is None:
                            done = True
                            break
                if done:
                    break
                if b.sampler:
                    x = self.conv(x, b.sampler, H, ups=True)
                    H *= 2
        if not done:
            n = self.groupnorm(x, None, "conv_norm_out", H * H, cfg.norm_eps, True)
            self.eps_out = self._alloc((B, cfg.out_channels, L, L))
            eo = self.eps_out
            self._add(lambda: ops.conv_out(n.t, w.h["conv_out.w"], w.f["conv_out.b"], B, L, out=eo))
        if self.grad:
            self._finalize_backward()

    def _finalize_backward(self):
        # Gradient fan-in is resolved statically: walking the ops in backward-execution order, the
        # first op that produces a gradient for a buffer overwrites it, later ones accumulate.
        seen = set()
        for op in reversed(self.ops):
            if op.make_bwd is None:
                continue
            op.acc = []
            for a in op.gouts:
                if a.g is None:
                    pt = op.passthrough
                    if pt is not None and a is pt[0] and pt[1].g is not None and pt[1].g.shape == a.t.shape \
                            and os.environ.get("LGD_RES_GRAD_ALIAS", "1") != "0":
                        a.g = pt[1].g                                # first writer of a pass-through gradient: no copy
                    else:
                        a.g = self._alloc(tuple(a.t.shape), a.t.dtype)   # first writer overwrites (acc flags below)
                op.acc.append(id(a) in seen)
                seen.add(id(a))
        # every consumed activation now owns a .g, so the closures can bind output gradients
        for op in self.ops:
            if op.make_bwd is not None:
                op.bwd = op.make_bwd(op.acc)
        self.g_latents = self._alloc(tuple(self.latents_in.shape))
        self._bwd_ops = [op.bwd for op in reversed(self.ops) if op.bwd is not None]

    # --------------------------------------------------------------------------------------
    def forward(self, latents: Optional[torch.Tensor] = None):
        """Enqueues the whole forward plan (graph-capturable: static pointers, no allocation)."""
        if latents is not None:
            self.latents_in.copy_(latents)
        for op in self.ops:
            op.fwd()
        return self.eps_out

    def backward(self, grad_scale: float = 1.0) -> torch.Tensor:
        """Runs the reverse plan; map gradients must already be in self.gmaps.  Returns the latent
        gradient (fp32 NCHW), divided by grad_scale."""
        for b in self._bwd_ops:
            b()
        eng = self.eng
        x0 = self._x0
        ops.conv_out(x0.g, eng.w.h["conv_in.wd"], None, self.B, self.L, out=self.g_latents,
                     out_scale=1.0 / grad_scale)
        return self.g_latents


class UNetEngine:
    ARENA_SEGMENT = 1 << 30      # bytes per arena segment (largest single plan buffer is ~0.2 GB)
    MAX_PLANS = 48               # LRU bound on cached launch plans (their buffers alias one arena anyway)

    def __init__(self, cfg: UNetConfig, device="cuda", state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 text_len: int = 77, max_text_batch: int = 32, weights: Optional[WeightStore] = None):
        """weights: the WeightStore of another engine of the same configuration on the same device — the two
        engines then read ONE copy of the parameters (read-only after load) while everything a run writes (time /
        text / GLIGEN tables, activation arena, split-K scratch, plans, graphs) stays per engine, so they can run
        concurrently on two HIP streams (lanes.LanePool)."""
        self.cfg = cfg
        self.device = torch.device(device)
        # GEMM tuning table of this engine's plans ("latency" | "throughput" | None = ops.current_tuning_mode() at
        # plan-build time); lanes.make_lanes sets "throughput" on the engines it hands to lanes.  Part of the plan key.
        self.tuning_mode: Optional[str] = None
        # LayerNorm folded into the consuming GEMM in no-grad plans (Plan.ln_linear); LGD_FOLD_LN=0 keeps the two ops
        self.fold_ln = os.environ.get("LGD_FOLD_LN", "1") != "0"
        self.blocks = unet_blocks(cfg)
        if weights is not None:
            if weights.cfg != cfg or torch.device(weights.device) != self.device or state_dict is not None:
                raise ValueError("shared weights must come from an engine of the same configuration and device")
            self.w = weights
            self.fold_ln = self.fold_ln and getattr(weights, "fold_ln", True)   # a store without the folded twins
        else:
            self.w = WeightStore(cfg, self.device, fold_ln=self.fold_ln)
            if state_dict is not None:
                self.w.load_state_dict(state_dict)
        self.text_len = text_len
        self.max_text_batch = max_text_batch
        # time-embedding projections of the current step: one row, or (text_time models) one row per image of the
        # conditioning batch — plans bind these addresses when they are built
        self.temb_rows = max_text_batch if cfg.addition_embed_type == "text_time" else 1
        self.temb_cur = torch.zeros((self.temb_rows, self.w.temb_total), device=self.device, dtype=F32)
        self.temb_table = None
        self.dyn = torch.zeros(4, device=self.device, dtype=torch.int32)   # {step, frozen_steps, -, -}
        self.step_idx = self.dyn[:1]
        self.text_kv: Dict[str, torch.Tensor] = {}
        for b in self.blocks:
            for a in b.attns:
                for dpt in range(a.depth):
                    self.text_kv[self.kv_name(a.prefix, dpt)] = torch.zeros((max_text_batch, text_len, 2 * a.channels),
                                                                            device=self.device, dtype=F16)
        self._ws = None
        self._ws_size = 0
        self._fuser_cat: Dict[Tuple, torch.Tensor] = {}
        self._plans: "OrderedDict[Tuple, Plan]" = OrderedDict()
        self._objs = None
        self._arena: List[torch.Tensor] = []      # uint8 segments shared by all plans (Plan._alloc)

    @staticmethod
    def kv_name(prefix: str, depth_index: int = 0) -> str:
        """Name of a transformer layer's text K/V (and GLIGEN concat) buffers: the attention block's prefix for its
        first layer (every SD 1.x / 2.x block has exactly one), "prefix@d" for layer d of a deeper block (SDXL)."""
        return prefix if depth_index == 0 else f"{prefix}@{depth_index}"

    # ---- activation arena -------------------------------------------------------------------
    def arena_take(self, cursor, shape, dtype):
        """Bump allocation of one plan buffer: `cursor` = [segment, byte offset] of the requesting plan.
        Segments are only ever appended (never moved or freed while the engine lives), so buffers handed
        out earlier — and the hipGraphs captured over them — stay valid when a larger plan arrives."""
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if nbytes > self.ARENA_SEGMENT:
            raise RuntimeError(f"plan buffer of {nbytes} bytes exceeds the arena segment size")
        seg, off = cursor
        off = (off + 255) & ~255
        if off + nbytes > self.ARENA_SEGMENT:
            seg, off = seg + 1, 0
        while len(self._arena) <= seg:
            self._arena.append(torch.empty(self.ARENA_SEGMENT, device=self.device, dtype=torch.uint8))
        t = self._arena[seg][off:off + nbytes].view(dtype).view(tuple(int(d) for d in shape))
        return t, [seg, off + nbytes]

    def arena_bytes(self) -> int:
        return len(self._arena) * self.ARENA_SEGMENT

    def poison_arena(self):
        """Debug/test aid: overwrite every plan buffer with NaN bit patterns — a plan that relied on data
        surviving another plan's run (or on build-time zeros) shows up as NaNs."""
        for seg in self._arena:
            seg.fill_(0xFF)

    # ---- shared scratch ---------------------------------------------------------------------
    def workspace(self, n_floats: int = 0) -> torch.Tensor:
        """One split-K workspace shared by all ops of THIS engine (they run back-to-back on one stream; another
        engine on another stream has its own)."""
        if self._ws is None:
            self._ws = torch.empty(ops.WS_FLOATS, device=self.device, dtype=F32)
        return self._ws

    def fuser_cat(self, prefix: str, B: int, S: int, text_off: int) -> torch.Tensor:
        key = (prefix, B, text_off)
        if key not in self._fuser_cat:
            C = self.text_kv[prefix].shape[-1] // 2
            self._fuser_cat[key] = torch.zeros((B * (S + N_OBJ_TOKENS), C), device=self.device, dtype=F16)
        return self._fuser_cat[key]

    # ---- per-run constants --------------------------------------------------------------------
    def prepare_timesteps(self, timesteps: Sequence[int], added_cond: Optional[Dict[str, torch.Tensor]] = None):
        """Time-embedding MLP + every resnet's time_emb_proj for all timesteps of the run
        (unet_2d_condition.py:785-808 + [ext] ResnetBlock2D): table [T][sum Cout] fp32.
        text_time models ([ext] diffusers SDXL UNet2DConditionModel.forward): added_cond = {"text_embeds": [R, pooled],
        "time_ids": [R, 5]} for the R images of the conditioning batch; emb_t + add_embedding(cat(text_embeds,
        Timesteps(time_ids))) is projected per image: table [T][R * sum Cout]."""
        self.const_writer = None          # whoever cached the previous tables must rebuild (dropin UNet wrapper)
        cfg, w = self.cfg, self.w
        c0 = cfg.block_out_channels[0]
        silu = torch.nn.functional.silu

        def sincos(v, dim):                                                       # flip_sin_to_cos, freq shift 0
            half = dim // 2
            freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=self.device) / half)
            e = v[:, None] * freqs[None]
            return torch.cat([torch.cos(e), torch.sin(e)], dim=-1)

        t = torch.as_tensor(list(timesteps), dtype=torch.float32, device=self.device)
        emb = sincos(t, c0).to(F16)
        T = emb.shape[0]
        h = ops.linear(emb.contiguous(), w.h["time_embedding.linear_1.w"], w.f["time_embedding.linear_1.b"], out_f32=True)
        h = silu(h).to(F16)
        h = ops.linear(h, w.h["time_embedding.linear_2.w"], w.f["time_embedding.linear_2.b"], out_f32=True)[:T]
        if cfg.addition_embed_type == "text_time":
            if added_cond is None:
                raise RuntimeError(f"{cfg.name} needs added_cond (text_embeds, time_ids) for its time embedding")
            te = added_cond["text_embeds"].to(self.device, F32)
            ids = added_cond["time_ids"].to(self.device, F32)
            R = te.shape[0]
            if R > self.temb_rows or te.shape[1] != cfg.pooled_dim or tuple(ids.shape) != (R, 5):
                raise RuntimeError(f"added_cond shapes {tuple(te.shape)} / {tuple(ids.shape)} do not fit {cfg.name}")
            tid = sincos(ids.reshape(-1), cfg.addition_time_embed_dim).reshape(R, -1)
            a = torch.cat([te, tid], dim=-1).to(F16).contiguous()
            a = ops.linear(a, w.h["add_embedding.linear_1.w"], w.f["add_embedding.linear_1.b"], out_f32=True)[:R]
            a = ops.linear(silu(a).to(F16), w.h["add_embedding.linear_2.w"], w.f["add_embedding.linear_2.b"], out_f32=True)[:R]
            h = (h[:, None, :] + a[None, :, :]).reshape(T * R, -1)                # emb = emb + aug_emb, per image
            tab = ops.linear(silu(h).to(F16).contiguous(), w.h["temb_proj.w"], w.f["temb_proj.b"], out_f32=True)[:T * R]
            full = torch.zeros((T, self.temb_rows, w.temb_total), device=self.device, dtype=F32)
            full[:, :R] = tab.view(T, R, -1)
            self.temb_table = full.view(T, -1)
        else:
            h = silu(h).to(F16)
            self.temb_table = ops.linear(h, w.h["temb_proj.w"], w.f["temb_proj.b"], out_f32=True)[:T].contiguous()
        return self.temb_table

    def set_step(self, index: int):
        """Selects the time-embedding row of step `index` (device-side copy, no kernel argument
        changes)."""
        self.step_idx.fill_(index)
        ops.select_row(self.temb_table, self.step_idx, self.temb_cur)

    def prepare_text(self, ehs: torch.Tensor):
        """to_k / to_v of all cross-attention layers for this prompt (time-invariant: computed once per
        run instead of once per UNet call; attention_processor.py:345-346,433-434)."""
        self.const_writer = None          # whoever cached the previous tables must rebuild (dropin UNet wrapper)
        Bt, T, Cx = ehs.shape
        if Bt > self.max_text_batch or T != self.text_len:
            raise RuntimeError(f"text batch {Bt}x{T} exceeds the engine's text K/V buffers "
                               f"({self.max_text_batch}x{self.text_len}); chunk the jobs (LMDSampler.max_batch)")
        x = ehs.to(self.device, F16).reshape(Bt * T, Cx).contiguous()
        for b in self.blocks:
            for a in b.attns:
                for dpt in range(a.depth):
                    kv = self.text_kv[self.kv_name(a.prefix, dpt)]
                    ops.linear(x, self.w.h[f"{a.prefix}.transformer_blocks.{dpt}.attn2.kv.w"],
                               out=kv.view(-1, kv.shape[-1])[:Bt * T])

    def prepare_gligen(self, boxes: torch.Tensor, masks: torch.Tensor, positive_embeddings: torch.Tensor):
        """GLIGEN grounding tokens for the run: PositionNet (unet_2d_condition.py:99-114), then per
        fuser layer linear(objs) and LayerNorm of those 30 rows, stored in the tail rows of the
        concat buffers (they do not depend on the latents or the timestep)."""
        self.const_writer = None          # whoever cached the previous tables must rebuild (dropin UNet wrapper)
        w = self.w
        dev = self.device
        boxes, masks, pe = boxes.to(dev, F32), masks.to(dev, F32), positive_embeddings.to(dev, F32)
        Bt = boxes.shape[0]
        m = masks.unsqueeze(-1)
        freq = (100 ** (torch.arange(8, device=dev) / 8)).float()
        xx = freq[None, None, None] * boxes.unsqueeze(-1)
        xyxy = torch.stack((xx.sin(), xx.cos()), dim=-1).permute(0, 1, 3, 4, 2).reshape(Bt, N_OBJ_TOKENS, -1)
        pos = pe * m + (1 - m) * w.f["position_net.null_positive_feature"].view(1, 1, -1)
        xyxy = xyxy * m + (1 - m) * w.f["position_net.null_position_feature"].view(1, 1, -1)
        h = torch.cat([pos, xyxy], dim=-1).reshape(Bt * N_OBJ_TOKENS, -1).to(F16).contiguous()
        silu = torch.nn.functional.silu
        h = silu(ops.linear(h, w.h["position_net.linears.0.w"], w.f["position_net.linears.0.b"], out_f32=True)).to(F16)
        h = silu(ops.linear(h, w.h["position_net.linears.2.w"], w.f["position_net.linears.2.b"], out_f32=True)).to(F16)
        objs = ops.linear(h, w.h["position_net.linears.4.w"], w.f["position_net.linears.4.b"])  # [Bt*30, Cx]
        self._objs = objs
        for (prefix, B, toff), cat in self._fuser_cat.items():
            if B not in (Bt, Bt // 2) or toff + B > Bt:
                continue                   # concat buffer of a plan with another batch size
            pfx, _, dpt = prefix.partition("@")                     # kv_name(prefix, depth index)
            f = f"{pfx}.transformer_blocks.{dpt or 0}.fuser"
            C = cat.shape[1]
            S = cat.shape[0] // B - N_OBJ_TOKENS
            o = ops.linear(objs, w.h[f"{f}.linear.w"], w.f[f"{f}.linear.b"])          # [Bt*30, C]
            o = ops.layernorm(o, w.f[f"{f}.norm1.g"], w.f[f"{f}.norm1.b"])
            cat.view(B, S + N_OBJ_TOKENS, C)[:, S:] = o.view(Bt, N_OBJ_TOKENS, C)[toff:toff + B]

    # ---- plans ---------------------------------------------------------------------------------
    def plan(self, B: int, L: int, *, grad=False, fuser=False, stop_key=None, save_keys=(),
             text_batch_offset=0, obj_batch_offset=0) -> Plan:
        mode = self.tuning_mode or ops.current_tuning_mode()
        key = (B, L, grad, fuser, tuple(stop_key) if stop_key else None, tuple(map(tuple, save_keys)),
               text_batch_offset, obj_batch_offset, mode, self.fold_ln)
        if B + text_batch_offset > self.max_text_batch:
            raise RuntimeError(f"plan batch {B} (+{text_batch_offset}) exceeds max_text_batch={self.max_text_batch}")
        if key not in self._plans:
            with ops.tuning(mode):                        # descriptors are tuned when they are built
                self._plans[key] = Plan(self, B, L, grad=grad, fuser=fuser, stop_key=stop_key,
                                        save_keys=save_keys, text_batch_offset=text_batch_offset,
                                        obj_batch_offset=obj_batch_offset)
            while len(self._plans) > self.MAX_PLANS:
                self._plans.popitem(last=False)       # a dropped plan's graphs stay valid: the arena never moves
        self._plans.move_to_end(key)
        return self._plans[key]

    def attn_key_order(self) -> List[Tuple]:
        """Attention keys in UNet execution order (down, mid, up)."""
        return [a.key for b in self.blocks for a in b.attns]

    def last_key(self, keys) -> Tuple:
        """The key of `keys` that executes last — where a guidance forward may stop (pipelines.py:46 TODO),
        whatever order the caller listed them in."""
        order = self.attn_key_order()
        return max((tuple(k) for k in keys), key=order.index)
