"""Device weight arenas for the HIP UNet engine.

All kernel-ready tensors live in two flat device buffers (fp16 matrices, fp32 vectors) so that the
data-parallel launcher can replicate a model with exactly two RCCL broadcasts over xGMI
 and no per-tensor traffic.  The arena layout is a pure function of the
UNetConfig, so non-root ranks allocate and slice their arenas without ever seeing a state dict.

Packing (host, once):
  conv 3x3   [Cout,Cin,3,3] -> [Cout][(ky,kx,ci)]            implicit-GEMM weight
  dgrad      -> [Cin][(ky',kx',co)] with taps flipped        same kernel computes the input gradient
  linear     [N,K] as is;  dgrad copy [K,N]
  attn1      to_q|to_k|to_v fused into one [3C,C] matrix (one GEMM, q/k/v are column views)
  attn2      to_k|to_v fused [2C,Cx] (text K/V are time-invariant: computed once per prompt)
  GEGLU      proj rows interleaved in 16-row (value, gate) blocks for the fused epilogue
  time_emb_proj of every resnet concatenated -> one [sum Cout, 4*C0] matrix (one GEMM per run)
"""
from typing import Dict, List, Tuple

import torch

from .weights import UNetConfig, unet_blocks

F16, F32 = torch.float16, torch.float32

# the guidance pass needs input-gradients only for layers at or before up_blocks[1].attentions[2]
# (pipelines.py:14 DEFAULT_GUIDANCE_ATTN_KEYS) — SURVEY.md §0.3


def _needs_bwd(prefix: str) -> bool:
    if prefix.startswith("up_blocks."):
        return int(prefix.split(".")[1]) <= 1
    return True


def geglu_perm(n: int) -> torch.Tensor:
    idx = []
    for j in range(n // 16):
        idx += list(range(16 * j, 16 * j + 16)) + list(range(n + 16 * j, n + 16 * j + 16))
    return torch.tensor(idx, dtype=torch.long)


def pack_conv(w: torch.Tensor) -> torch.Tensor:
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_conv_dgrad(w: torch.Tensor) -> torch.Tensor:
    # Wd[ci][(ky',kx',co)] = W[co][ci][2-ky'][2-kx']
    return pack_conv(w.flip(2, 3).permute(1, 0, 2, 3).contiguous())


class WeightStore:
    def __init__(self, cfg: UNetConfig, device, fold_ln: bool = True):
        """fold_ln: also lay out (and fill) the LayerNorm-folded twins of the linears behind a LayerNorm (`_lnlin`: 12 of
        a transformer block's 20 linear weights a second time, 0.5 GB for SD1.4+GLIGEN, ~1 GB for the SDXL refiner) —
        the no-grad plans of an engine with `fold_ln` read them; an engine built with LGD_FOLD_LN=0 neither stores nor
        broadcasts them."""
        self.fold_ln = bool(fold_ln)
        if cfg.in_channels > 4:
            raise ValueError(f"in_channels = {cfg.in_channels}: conv_in runs as an implicit GEMM over an 8-channel copy of the "
                             "latents (value + fp16 rounding remainder of <= 4 channels); inpainting UNets are not supported")
        self.cfg = cfg
        self.device = device
        self.blocks = unet_blocks(cfg)
        self.entries16: List[Tuple[str, Tuple[int, ...]]] = []
        self.entries32: List[Tuple[str, Tuple[int, ...]]] = []
        self.temb_offsets: Dict[str, int] = {}
        self._layout()
        n16 = sum(self._numel(s) for _, s in self.entries16)
        n32 = sum(self._numel(s) for _, s in self.entries32)
        self.arena16 = torch.zeros(n16, device=device, dtype=F16)
        self.arena32 = torch.zeros(n32, device=device, dtype=F32)
        self.h: Dict[str, torch.Tensor] = {}
        self.f: Dict[str, torch.Tensor] = {}
        off = 0
        for k, s in self.entries16:
            self.h[k] = self.arena16[off:off + self._real(s)].view(s)
            off += self._numel(s)
        off = 0
        for k, s in self.entries32:
            self.f[k] = self.arena32[off:off + self._real(s)].view(s)
            off += self._numel(s)
        self.scalars: Dict[str, float] = {}   # tanh(alpha) gates (host floats; tiny)

    @staticmethod
    def _real(s):
        n = 1
        for d in s:
            n *= d
        return n

    @staticmethod
    def _numel(s):
        return (WeightStore._real(s) + 7) // 8 * 8      # keep every tensor 16-byte aligned

    # ------------------------------------------------------------------------------------------
    def _e16(self, k, s):
        self.entries16.append((k, tuple(s)))

    def _e32(self, k, s):
        self.entries32.append((k, tuple(s)))

    def _lin(self, name, n, k, bias=True, bwd=False):
        self._e16(f"{name}.w", (n, k))
        if bwd:
            self._e16(f"{name}.wt", (k, n))
        if bias:
            self._e32(f"{name}.b", (n,))

    def _lnlin(self, name, n, k):
        """The LayerNorm-folded twin of a linear layer whose input is a LayerNorm (no-grad plans; LGD_EPI_ROWNORM in
        include/lgd_hip.h): wln = W * gamma per input channel, cs = row sums of wln AS STORED (fp16), bln = b + W beta."""
        if not self.fold_ln:
            return
        self._e16(f"{name}.wln", (n, k))
        self._e32(f"{name}.cs", (n,))
        self._e32(f"{name}.bln", (n,))

    def _conv(self, name, cout, cin, bwd=False):
        self._e16(f"{name}.w", (cout, 9 * cin))
        if bwd:
            self._e16(f"{name}.wd", (cin, 9 * cout))
        self._e32(f"{name}.b", (cout,))

    def _normp(self, name, c):
        self._e32(f"{name}.g", (c,))
        self._e32(f"{name}.b", (c,))

    def _layout(self):
        cfg = self.cfg
        c0, ted, cx = cfg.block_out_channels[0], cfg.time_embed_dim, cfg.cross_attention_dim
        self._e16("conv_in.w", (c0, 9 * cfg.in_channels))
        self._e16("conv_in.wd", (cfg.in_channels, 9 * c0))
        self._e16("conv_in.w8", (c0, 72))             # the same filter over 8 input channels (4..7 zero): implicit-GEMM form
        self._e32("conv_in.b", (c0,))
        self._lin("time_embedding.linear_1", ted, c0)
        self._lin("time_embedding.linear_2", ted, ted)
        if cfg.addition_embed_type == "text_time":
            self._lin("add_embedding.linear_1", ted, cfg.projection_class_embeddings_input_dim)
            self._lin("add_embedding.linear_2", ted, ted)
        toff = 0
        for b in self.blocks:
            for r in b.resnets:
                bw = _needs_bwd(r.prefix)
                self._normp(f"{r.prefix}.norm1", r.cin)
                self._conv(f"{r.prefix}.conv1", r.cout, r.cin, bw)
                self._normp(f"{r.prefix}.norm2", r.cout)
                self._conv(f"{r.prefix}.conv2", r.cout, r.cout, bw)
                if r.shortcut:
                    self._lin(f"{r.prefix}.conv_shortcut", r.cout, r.cin, bwd=bw)
                self.temb_offsets[r.prefix] = toff
                toff += r.cout
            for a in b.attns:
                bw = _needs_bwd(a.prefix)
                C = a.channels
                self._normp(f"{a.prefix}.norm", C)
                self._lin(f"{a.prefix}.proj_in", C, C, bwd=bw)
                for d in range(a.depth):
                    t = f"{a.prefix}.transformer_blocks.{d}"
                    self._normp(f"{t}.norm1", C)
                    self._lin(f"{t}.attn1.qkv", 3 * C, C, bias=False, bwd=bw)
                    self._lnlin(f"{t}.attn1.qkv", 3 * C, C)
                    self._lin(f"{t}.attn1.to_out.0", C, C, bwd=bw)
                    self._normp(f"{t}.norm2", C)
                    self._lin(f"{t}.attn2.to_q", C, C, bias=False, bwd=bw)
                    self._lnlin(f"{t}.attn2.to_q", C, C)
                    self._lin(f"{t}.attn2.kv", 2 * C, cx, bias=False)
                    self._lin(f"{t}.attn2.to_out.0", C, C, bwd=bw)
                    self._normp(f"{t}.norm3", C)
                    self._lin(f"{t}.ff.net.0.proj", 8 * C, C, bwd=bw)
                    self._lnlin(f"{t}.ff.net.0.proj", 8 * C, C)
                    self._lin(f"{t}.ff.net.2", C, 4 * C, bwd=bw)
                    if cfg.use_gated_attention:
                        f = f"{t}.fuser"
                        self._lin(f"{f}.linear", C, cx)
                        self._normp(f"{f}.norm1", C)
                        self._lin(f"{f}.attn.qkv", 3 * C, C, bias=False, bwd=bw)
                        self._lin(f"{f}.attn.to_out.0", C, C, bwd=bw)
                        self._normp(f"{f}.norm2", C)
                        self._lin(f"{f}.ff.net.0.proj", 8 * C, C, bwd=bw)
                        self._lnlin(f"{f}.ff.net.0.proj", 8 * C, C)
                        self._lin(f"{f}.ff.net.2", C, 4 * C, bwd=bw)
                self._lin(f"{a.prefix}.proj_out", C, C, bwd=bw)
            if b.sampler:
                self._conv(b.sampler, b.channels, b.channels, _needs_bwd(b.sampler))
        self.temb_total = toff
        self._lin("temb_proj", toff, ted)
        self._normp("conv_norm_out", c0)
        self._e16("conv_out.w", (cfg.out_channels, 9 * c0))
        self._e32("conv_out.b", (cfg.out_channels,))
        if cfg.use_gated_attention:
            self._lin("position_net.linears.0", 512, cfg.gligen_positive_len + 64)
            self._lin("position_net.linears.2", 512, 512)
            self._lin("position_net.linears.4", cx, 512)
            self._e32("position_net.null_positive_feature", (cfg.gligen_positive_len,))
            self._e32("position_net.null_position_feature", (64,))
            self._e32("fuser_gates", (2 * sum(a.depth for b in self.blocks for a in b.attns),))

    # ------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Packs a reference-named fp32 state dict (CPU) into the arenas."""
        cfg = self.cfg
        h16: Dict[str, torch.Tensor] = {}
        f32: Dict[str, torch.Tensor] = {}

        def lin(dst, src, bias=True, bwd=False, w=None, b=None):
            w = sd[f"{src}.weight"] if w is None else w
            if w.dim() == 4:
                w = w.reshape(w.shape[0], w.shape[1])
            h16[f"{dst}.w"] = w
            if bwd:
                h16[f"{dst}.wt"] = w.t().contiguous()
            if bias:
                f32[f"{dst}.b"] = sd[f"{src}.bias"] if b is None else b

        def conv(name, bwd):
            w = sd[f"{name}.weight"]
            h16[f"{name}.w"] = pack_conv(w)
            if bwd:
                h16[f"{name}.wd"] = pack_conv_dgrad(w)
            f32[f"{name}.b"] = sd[f"{name}.bias"]

        def normp(name):
            f32[f"{name}.g"] = sd[f"{name}.weight"]
            f32[f"{name}.b"] = sd[f"{name}.bias"]

        def geglu(dst, src, bwd):
            w, b = sd[f"{src}.weight"], sd[f"{src}.bias"]
            perm = geglu_perm(w.shape[0] // 2)
            lin(dst, None, bwd=bwd, w=w[perm].contiguous(), b=b[perm].contiguous())

        def fold(dst, norm):
            # after lin()/geglu(): the rows are already in kernel order, the fold runs along K
            if not self.fold_ln:
                return
            w = h16[f"{dst}.w"].to(F16).float()                      # the values the unfolded GEMM multiplies by
            g, bt = sd[f"{norm}.weight"].float(), sd[f"{norm}.bias"].float()
            wln = (w * g[None, :]).to(F16)
            h16[f"{dst}.wln"] = wln
            f32[f"{dst}.cs"] = wln.float().sum(dim=1)
            b0 = f32.get(f"{dst}.b")
            f32[f"{dst}.bln"] = w @ bt + (b0.float() if b0 is not None else 0.0)

        h16["conv_in.w"] = pack_conv(sd["conv_in.weight"])
        h16["conv_in.wd"] = pack_conv_dgrad(sd["conv_in.weight"])
        wi = sd["conv_in.weight"]
        ci = wi.shape[1]
        parts = [wi, wi] if 2 * ci <= 8 else [wi]            # second copy multiplies the fp16 rounding remainder of the latents
        parts.append(wi.new_zeros(wi.shape[0], 8 - ci * len(parts), 3, 3))
        h16["conv_in.w8"] = pack_conv(torch.cat(parts, dim=1))
        f32["conv_in.b"] = sd["conv_in.bias"]
        lin("time_embedding.linear_1", "time_embedding.linear_1")
        lin("time_embedding.linear_2", "time_embedding.linear_2")
        if cfg.addition_embed_type == "text_time":
            lin("add_embedding.linear_1", "add_embedding.linear_1")
            lin("add_embedding.linear_2", "add_embedding.linear_2")
        tw, tb = [], []
        gates = []
        for b in self.blocks:
            for r in b.resnets:
                bw = _needs_bwd(r.prefix)
                normp(f"{r.prefix}.norm1")
                conv(f"{r.prefix}.conv1", bw)
                normp(f"{r.prefix}.norm2")
                conv(f"{r.prefix}.conv2", bw)
                if r.shortcut:
                    lin(f"{r.prefix}.conv_shortcut", f"{r.prefix}.conv_shortcut", bwd=bw)
                tw.append(sd[f"{r.prefix}.time_emb_proj.weight"])
                tb.append(sd[f"{r.prefix}.time_emb_proj.bias"])
            for a in b.attns:
                bw = _needs_bwd(a.prefix)
                normp(f"{a.prefix}.norm")
                lin(f"{a.prefix}.proj_in", f"{a.prefix}.proj_in", bwd=bw)
                for d in range(a.depth):
                    t = f"{a.prefix}.transformer_blocks.{d}"
                    normp(f"{t}.norm1")
                    qkv = torch.cat([sd[f"{t}.attn1.to_q.weight"], sd[f"{t}.attn1.to_k.weight"],
                                     sd[f"{t}.attn1.to_v.weight"]], dim=0)
                    lin(f"{t}.attn1.qkv", None, bias=False, bwd=bw, w=qkv)
                    fold(f"{t}.attn1.qkv", f"{t}.norm1")
                    lin(f"{t}.attn1.to_out.0", f"{t}.attn1.to_out.0", bwd=bw)
                    normp(f"{t}.norm2")
                    lin(f"{t}.attn2.to_q", f"{t}.attn2.to_q", bias=False, bwd=bw)
                    fold(f"{t}.attn2.to_q", f"{t}.norm2")
                    kv = torch.cat([sd[f"{t}.attn2.to_k.weight"], sd[f"{t}.attn2.to_v.weight"]], dim=0)
                    lin(f"{t}.attn2.kv", None, bias=False, w=kv)
                    lin(f"{t}.attn2.to_out.0", f"{t}.attn2.to_out.0", bwd=bw)
                    normp(f"{t}.norm3")
                    geglu(f"{t}.ff.net.0.proj", f"{t}.ff.net.0.proj", bw)
                    fold(f"{t}.ff.net.0.proj", f"{t}.norm3")
                    lin(f"{t}.ff.net.2", f"{t}.ff.net.2", bwd=bw)
                    if cfg.use_gated_attention:
                        f = f"{t}.fuser"
                        lin(f"{f}.linear", f"{f}.linear")
                        normp(f"{f}.norm1")
                        qkv = torch.cat([sd[f"{f}.attn.to_q.weight"], sd[f"{f}.attn.to_k.weight"],
                                         sd[f"{f}.attn.to_v.weight"]], dim=0)
                        lin(f"{f}.attn.qkv", None, bias=False, bwd=bw, w=qkv)
                        lin(f"{f}.attn.to_out.0", f"{f}.attn.to_out.0", bwd=bw)
                        normp(f"{f}.norm2")
                        geglu(f"{f}.ff.net.0.proj", f"{f}.ff.net.0.proj", bw)
                        fold(f"{f}.ff.net.0.proj", f"{f}.norm2")
                        lin(f"{f}.ff.net.2", f"{f}.ff.net.2", bwd=bw)
                        gates += [float(sd[f"{f}.alpha_attn"].tanh()), float(sd[f"{f}.alpha_dense"].tanh())]
                lin(f"{a.prefix}.proj_out", f"{a.prefix}.proj_out", bwd=bw)
            if b.sampler:
                conv(b.sampler, _needs_bwd(b.sampler))
        lin("temb_proj", None, w=torch.cat(tw, dim=0), b=torch.cat(tb, dim=0))
        normp("conv_norm_out")
        h16["conv_out.w"] = pack_conv(sd["conv_out.weight"])
        f32["conv_out.b"] = sd["conv_out.bias"]
        if cfg.use_gated_attention:
            for i in (0, 2, 4):
                lin(f"position_net.linears.{i}", f"position_net.linears.{i}")
            f32["position_net.null_positive_feature"] = sd["position_net.null_positive_feature"]
            f32["position_net.null_position_feature"] = sd["position_net.null_position_feature"]
            f32["fuser_gates"] = torch.tensor(gates, dtype=F32)
        # host-side flat packing, then one H2D copy per arena
        flat16 = torch.zeros(self.arena16.numel(), dtype=F16)
        off = 0
        for k, s in self.entries16:
            n = self._numel(s)
            t = h16[k]
            assert tuple(t.shape) == s, (k, tuple(t.shape), s)
            flat16[off:off + t.numel()] = t.reshape(-1).to(F16)
            off += n
        flat32 = torch.zeros(self.arena32.numel(), dtype=F32)
        off = 0
        for k, s in self.entries32:
            n = self._numel(s)
            t = f32[k]
            assert tuple(t.shape) == s, (k, tuple(t.shape), s)
            flat32[off:off + t.numel()] = t.reshape(-1).float()
            off += n
        self.arena16.copy_(flat16)
        self.arena32.copy_(flat32)
        self.refresh_scalars()

    def refresh_scalars(self):
        """Host copies of the tanh(alpha) fuser gates (after load or after an RCCL broadcast)."""
        if self.cfg.use_gated_attention:
            g = self.f["fuser_gates"].cpu().tolist()
            i = 0
            for b in self.blocks:
                for a in b.attns:
                    for d in range(a.depth):
                        t = f"{a.prefix}.transformer_blocks.{d}.fuser"
                        self.scalars[f"{t}.alpha_attn"] = g[i]
                        self.scalars[f"{t}.alpha_dense"] = g[i + 1]
                        i += 2

    def nbytes(self):
        return self.arena16.numel() * 2 + self.arena32.numel() * 4
