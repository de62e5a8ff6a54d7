"""VAE decoder ([ext] diffusers AutoencoderKL, SD 1.x / 2.x config).  It sits inside the images/s window
because pipelines.decode (models/pipelines.py:117-127, 588-595) runs once per box and once per image
(SURVEY.md §8a P7, §8f rank 1).

  * HipVAEDecoder   — runs an AutoencoderKL DECODER state dict (`AutoencoderKL.state_dict()` of a real checkpoint,
                      models/models.py:41, or the seeded `synth_aekl_state_dict()`) on the engine's own kernels:
                      implicit-GEMM convolutions, GroupNorm+SiLU, and for the single 512-wide mid-block attention
                      three batched GEMM launches + a row softmax for the whole batch.
  * the plain PyTorch fp32 restatement the tests compare it with lives in oracle/restate_vae.py (test infrastructure).

Parity note: AutoencoderKL itself (diffusers 0.18.0) is absent from the sandbox, so the torch module below is a
restatement from the published architecture — "parity unpinned" at that boundary; what the tests pin is
HipVAEDecoder against the fp32 restatement of oracle/restate_vae.py at the full SD size, through the AutoencoderKL key map.

Architecture (SD config: block_out_channels (128,256,512,512), layers_per_block 2, 32 groups, eps 1e-6):
post_quant_conv 4->4 (1x1), conv_in 4->512, mid (resnet, 1-head attention, resnet), 4 up blocks
(512,512,256,128) x 3 resnets with 3 nearest-2x upsamplers, GroupNorm+SiLU, conv_out 128->3.
"""
import re

import torch


class HipVAEDecoder:
    """AutoencoderKL.decode (the `.sample` tensor) on the lgd_hip kernels: channels-last fp16, fp32 accumulate.

    HipVAEDecoder(state_dict | module with `aekl_state_dict()`, device).  The state dict is AutoencoderKL's (encoder keys are ignored);
    the decoder's shape (channels per up block, resnets per block, which blocks upsample) is read off the keys."""

    def __init__(self, source, device, groups: int = 32, eps: float = 1e-6):
        from . import ops
        from .weightstore import pack_conv
        self.ops = ops
        self.dev = torch.device(device)
        self.groups, self.eps = groups, eps
        sd = source.aekl_state_dict() if hasattr(source, "aekl_state_dict") else dict(source)
        sd = {k: v.detach().float().cpu() for k, v in sd.items() if k.startswith(("decoder.", "post_quant_conv."))}
        if "decoder.conv_in.weight" not in sd:
            raise RuntimeError("not an AutoencoderKL state dict: decoder.conv_in.weight is missing")
        h16 = lambda t: t.to(self.dev, torch.float16).contiguous()
        f32 = lambda t: t.to(self.dev, torch.float32).contiguous()
        conv = lambda n: (h16(pack_conv(sd[f"{n}.weight"])), f32(sd[f"{n}.bias"]))
        lin = lambda n: (h16(sd[f"{n}.weight"].reshape(sd[f"{n}.weight"].shape[0], -1)), f32(sd[f"{n}.bias"]))
        norm = lambda n: (f32(sd[f"{n}.weight"]), f32(sd[f"{n}.bias"]))

        def res(n):
            return dict(n1=norm(f"{n}.norm1"), c1=conv(f"{n}.conv1"), n2=norm(f"{n}.norm2"), c2=conv(f"{n}.conv2"),
                        sc=lin(f"{n}.conv_shortcut") if f"{n}.conv_shortcut.weight" in sd else None)

        # post_quant_conv (1x1, 4 -> 4) folded into conv_in: conv_in(W1 z + b1) = conv_in'(z) + border-aware bias map
        # (the zero padding of conv_in does not carry b1, so the folded bias depends on which taps lie inside the image)
        w_in, b_in = sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"]
        if "post_quant_conv.weight" in sd:
            w1 = sd["post_quant_conv.weight"].reshape(sd["post_quant_conv.weight"].shape[0], -1)
            b1 = sd["post_quant_conv.bias"]
            self._tap_bias = torch.einsum("ockl,c->okl", w_in, b1)                   # [Cout,3,3]
            w_in = torch.einsum("ockl,cd->odkl", w_in, w1)
        else:
            self._tap_bias = torch.zeros(w_in.shape[0], 3, 3)
        self._b_in = b_in
        ci = w_in.shape[1]
        if ci > 4:
            raise RuntimeError(f"latent_channels = {ci}: the 8-channel conv_in operand holds value + remainder of <= 4 channels")
        # 8-channel operand of lgd_nchw_to_nhwc8_f16: channels 0..ci-1 = fp16 value, ci..2ci-1 = its rounding remainder
        self.conv_in_w8 = h16(pack_conv(torch.cat([w_in, w_in, w_in.new_zeros(w_in.shape[0], 8 - 2 * ci, 3, 3)], dim=1)))
        self.c_mid = w_in.shape[0]
        self._bias_maps = {}

        a = "decoder.mid_block.attentions.0"
        legacy = f"{a}.query.weight" in sd
        nq, nk, nv, no = ("query", "key", "value", "proj_attn") if legacy else ("to_q", "to_k", "to_v", "to_out.0")
        # the softmax scale C^-1/2 is split over the q and k projections, so the fp16 score matrix holds scaled logits
        # (raw 512-term dot products of a real VAE can leave the fp16 range)
        s4 = float(self.c_mid) ** -0.25
        wq, wk = sd[f"{a}.{nq}.weight"].reshape(self.c_mid, -1) * s4, sd[f"{a}.{nk}.weight"].reshape(self.c_mid, -1) * s4
        self.attn = dict(n=norm(f"{a}.group_norm"),
                         qk=(h16(torch.cat([wq, wk])), f32(torch.cat([sd[f"{a}.{nq}.bias"], sd[f"{a}.{nk}.bias"]]) * s4)),
                         v=h16(sd[f"{a}.{nv}.weight"].reshape(self.c_mid, -1)), vb=f32(sd[f"{a}.{nv}.bias"]),
                         o=(h16(sd[f"{a}.{no}.weight"].reshape(self.c_mid, -1)), f32(sd[f"{a}.{no}.bias"])))
        self.mid = [res("decoder.mid_block.resnets.0"), res("decoder.mid_block.resnets.1")]
        n_up = 1 + max(int(m.group(1)) for k in sd for m in [re.match(r"decoder\.up_blocks\.(\d+)\.", k)] if m)
        self.ups = []
        for i in range(n_up):
            n_res = 1 + max(int(m.group(1)) for k in sd
                            for m in [re.match(rf"decoder\.up_blocks\.{i}\.resnets\.(\d+)\.", k)] if m)
            up = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            self.ups.append(([res(f"decoder.up_blocks.{i}.resnets.{j}") for j in range(n_res)],
                             conv(up) if f"{up}.weight" in sd else None))
        self.norm_out = norm("decoder.conv_norm_out")
        w = pack_conv(sd["decoder.conv_out.weight"])                                 # [3, 9*C]
        self.conv_out = (h16(torch.cat([w, torch.zeros(1, w.shape[1])])),              # pad to 4 output channels
                         f32(torch.cat([sd["decoder.conv_out.bias"], torch.zeros(1)])))

    @classmethod
    def from_state_dict(cls, state_dict, device, **kw):
        """`AutoencoderKL.state_dict()` (models/models.py:41) -> decoder on the HIP kernels."""
        return cls(state_dict, device, **kw)

    def _bias_map(self, B, L):
        """conv_in bias + what post_quant_conv's bias contributes through the taps that lie inside the image:
        fp16 [B*L*L, C] (residual operand of the conv_in GEMM)."""
        key = (B, L)
        if key not in self._bias_maps:
            v = torch.ones(3, L)
            v[0, 0] = 0          # tap row ky = 0 reads y - 1: outside at y = 0
            v[2, L - 1] = 0
            m = torch.einsum("okl,ky,lx->yxo", self._tap_bias, v, v) + self._b_in            # [L, L, C]
            self._bias_maps = {key: m.reshape(1, L * L, -1).expand(B, -1, -1).reshape(B * L * L, -1)
                               .to(self.dev, torch.float16).contiguous()}
        return self._bias_maps[key]

    def _res(self, p, x, B, H):
        ops = self.ops
        HW = H * H
        h = ops.groupnorm(x, B, HW, self.groups, self.eps, p["n1"][0], p["n1"][1], True)
        h = ops.conv3x3(h, p["c1"][0], B, H, H, bias=p["c1"][1])
        h = ops.groupnorm(h, B, HW, self.groups, self.eps, p["n2"][0], p["n2"][1], True)
        sc = x if p["sc"] is None else ops.linear(x, p["sc"][0], p["sc"][1])
        return ops.conv3x3(h, p["c2"][0], B, H, H, bias=p["c2"][1], res=sc)

    def _attn(self, p, x, B, H):
        """Single-head attention over the H*H positions at width C (512): the head is wider than the flash kernels'
        160, so the batch runs as three BATCHED GEMM launches + one row softmax:
            scores[b] = q[b] k[b]^T      ->  P = softmax(scores)      (q, k carry C^-1/4 each)
            vt[b]     = Wv n[b]^T        (V^T directly from the projection: no transpose pass)
            out[b]    = P[b] vt[b]^T + bv   (rows of P sum to 1, so V's bias is added once, behind the product)"""
        ops = self.ops
        S = H * H
        C = x.shape[1]
        F16 = torch.float16
        n = ops.groupnorm(x, B, S, self.groups, self.eps, p["n"][0], p["n"][1], False)
        qk = ops.linear(n, p["qk"][0], p["qk"][1])                                  # [B*S, 2C]
        sc = torch.empty((B * S, S), device=self.dev, dtype=F16)
        ops.gemm_launch(ops.gemm_desc(qk, qk[:, C:], sc, S, S, C, lda0=2 * C, ldw=2 * C, ldc=S, nb_o=B,
                                      a_bs=(S * 2 * C, 0), w_bs=(S * 2 * C, 0), c_bs=(S * S, 0)))
        pr = ops.softmax_rows(sc, 1.0, out=sc)
        vt = torch.empty((B * C, S), device=self.dev, dtype=F16)
        ops.gemm_launch(ops.gemm_desc(p["v"], n, vt, C, S, C, lda0=C, ldw=C, ldc=S, nb_o=B,
                                      a_bs=(0, 0), w_bs=(S * C, 0), c_bs=(C * S, 0)))
        a = torch.empty((B * S, C), device=self.dev, dtype=F16)
        ops.gemm_launch(ops.gemm_desc(pr, vt, a, S, C, S, lda0=S, ldw=S, ldc=C, bias=p["vb"], nb_o=B,
                                      a_bs=(S * S, 0), w_bs=(C * S, 0), c_bs=(S * C, 0)))
        return ops.linear(a, p["o"][0], p["o"][1], res=x)

    @torch.no_grad()
    def decode(self, z):
        ops = self.ops
        z = z.to(self.dev, torch.float32).contiguous()
        B, _, L, _ = z.shape
        lat8 = ops.nchw_to_nhwc8(z)
        h = torch.empty((B * L * L, self.c_mid), device=self.dev, dtype=torch.float16)
        ops.gemm_launch(ops.gemm_desc(lat8, self.conv_in_w8, h, B * L * L, self.c_mid, 72, c0=8, lda0=8, taps=9, hin=L,
                                      win=L, hout=L, wout=L, res=self._bias_map(B, L), ldr=self.c_mid, ldc=self.c_mid,
                                      splits=1))
        H = L
        h = self._res(self.mid[0], h, B, H)
        h = self._attn(self.attn, h, B, H)
        h = self._res(self.mid[1], h, B, H)
        for blk, up in self.ups:
            for r in blk:
                h = self._res(r, h, B, H)
            if up is not None:
                h = ops.conv3x3(h, up[0], B, H, H, bias=up[1], ups=1)
                H *= 2
        h = ops.groupnorm(h, B, H * H, self.groups, self.eps, self.norm_out[0], self.norm_out[1], True)
        # conv_out (128 -> 3, padded to 4 channels) on the matrix cores as well: at 512 x 512 the one-wave-per-pixel
        # kernel took 1.7 ms per image, the implicit GEMM (N = 4 inside a 64-wide tile) a few tens of microseconds
        y = ops.conv3x3(h, self.conv_out[0], B, H, H, bias=self.conv_out[1])            # [B*H*H, 4] fp16
        return y.view(B, H, H, 4)[..., :3].permute(0, 3, 1, 2).float()


def make_hip_vae(device, seed=0):
    """Random-init SD VAE decoder (seeded AutoencoderKL state dict of the SD architecture) running on the HIP kernels."""
    return HipVAEDecoder(synth_aekl_state_dict(seed=seed), device)


def synth_aekl_state_dict(ch=(128, 256, 512, 512), layers=2, latent_channels=4, seed=0, legacy_attention_names=False):
    """A full AutoencoderKL state dict (encoder + quant_conv + post_quant_conv + decoder, diffusers key names) with
    seeded random parameters of the SD / SDXL VAE architecture: there are no checkpoints in the sandbox."""
    g = torch.Generator().manual_seed(4321 + seed)
    out = {}

    def conv(name, cout, cin, k):
        out[f"{name}.weight"] = (torch.rand((cout, cin, k, k), generator=g) * 2 - 1) * (cin * k * k) ** -0.5
        out[f"{name}.bias"] = 0.05 * torch.randn((cout,), generator=g)

    def norm(name, c):
        out[f"{name}.weight"] = 1.0 + 0.1 * torch.randn((c,), generator=g)
        out[f"{name}.bias"] = 0.05 * torch.randn((c,), generator=g)

    def res(name, cin, cout):
        norm(f"{name}.norm1", cin); conv(f"{name}.conv1", cout, cin, 3)
        norm(f"{name}.norm2", cout); conv(f"{name}.conv2", cout, cout, 3)
        if cin != cout:
            conv(f"{name}.conv_shortcut", cout, cin, 1)

    def attn(name, c):
        norm(f"{name}.group_norm", c)
        for n in (("query", "key", "value", "proj_attn") if legacy_attention_names else ("to_q", "to_k", "to_v", "to_out.0")):
            out[f"{name}.{n}.weight"] = (torch.rand((c, c), generator=g) * 2 - 1) * c ** -0.5
            out[f"{name}.{n}.bias"] = 0.05 * torch.randn((c,), generator=g)

    conv("encoder.conv_in", ch[0], 3, 3)
    cin = ch[0]
    for i, c in enumerate(ch):
        for j in range(layers):
            res(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else c, c)
        cin = c
        if i < len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
    res("encoder.mid_block.resnets.0", cin, cin); attn("encoder.mid_block.attentions.0", cin); res("encoder.mid_block.resnets.1", cin, cin)
    norm("encoder.conv_norm_out", cin); conv("encoder.conv_out", 2 * latent_channels, cin, 3)
    conv("quant_conv", 2 * latent_channels, 2 * latent_channels, 1)
    conv("post_quant_conv", latent_channels, latent_channels, 1)
    rev = tuple(reversed(ch))
    conv("decoder.conv_in", rev[0], latent_channels, 3)
    res("decoder.mid_block.resnets.0", rev[0], rev[0]); attn("decoder.mid_block.attentions.0", rev[0]); res("decoder.mid_block.resnets.1", rev[0], rev[0])
    cin = rev[0]
    for i, c in enumerate(rev):
        for j in range(layers + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else c, c)
        cin = c
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    norm("decoder.conv_norm_out", cin); conv("decoder.conv_out", 3, cin, 3)
    return out


def fold_pointwise_after(w, b, w1, b1):
    """A 1x1 convolution (w1 [P, C, 1, 1], b1) applied AFTER a k x k convolution (w [C, D, k, k], b) is one k x k
    convolution: weights sum_c w1[p, c] w[c, d, ky, kx], bias w1 b + b1 (exact: no padding interaction on this side)."""
    m = w1.reshape(w1.shape[0], -1)
    return torch.einsum("pc,cdkl->pdkl", m, w), m @ b + b1


class HipVAEEncoder(HipVAEDecoder):
    """AutoencoderKL.encode(image).latent_dist on the lgd_hip kernels — the first half of the SDXL-refiner img2img
    pass (generation/sdxl_refinement.py:29 -> [ext] StableDiffusionXLImg2ImgPipeline.prepare_latents): conv_in 3 -> C0,
    down blocks (resnets without time embedding, Downsample2D = pad right/bottom by one + 3x3 stride 2), mid block
    (resnet, one-head attention, resnet), GroupNorm + SiLU, conv_out -> 2 x latent channels, quant_conv (1x1).

    Downsample2D's asymmetric padding: the implicit-GEMM gather pads symmetrically, so the stride-2 output is taken
    from the ODD positions of the stride-1 'same' convolution (out[o] = sum_k x[2o + k] w[k] = same[2o + 1]; the
    position past the border reads the zero the asymmetric pad adds) — 4x the flops of three small convolutions,
    once per image."""

    def __init__(self, state_dict, device, groups: int = 32, eps: float = 1e-6):
        from . import ops
        from .weightstore import pack_conv
        self.ops = ops
        self.dev = torch.device(device)
        self.groups, self.eps = groups, eps
        sd = {k: v.detach().float().cpu() for k, v in dict(state_dict).items() if k.startswith(("encoder.", "quant_conv."))}
        if "encoder.conv_in.weight" not in sd:
            raise RuntimeError("not an AutoencoderKL state dict: encoder.conv_in.weight is missing")
        h16 = lambda t: t.to(self.dev, torch.float16).contiguous()
        f32 = lambda t: t.to(self.dev, torch.float32).contiguous()
        conv = lambda n: (h16(pack_conv(sd[f"{n}.weight"])), f32(sd[f"{n}.bias"]))
        lin = lambda n: (h16(sd[f"{n}.weight"].reshape(sd[f"{n}.weight"].shape[0], -1)), f32(sd[f"{n}.bias"]))
        norm = lambda n: (f32(sd[f"{n}.weight"]), f32(sd[f"{n}.bias"]))

        def res(n):
            return dict(n1=norm(f"{n}.norm1"), c1=conv(f"{n}.conv1"), n2=norm(f"{n}.norm2"), c2=conv(f"{n}.conv2"),
                        sc=lin(f"{n}.conv_shortcut") if f"{n}.conv_shortcut.weight" in sd else None)
        w_in = sd["encoder.conv_in.weight"]                                          # [C0, 3, 3, 3]
        ci = w_in.shape[1]
        if 2 * ci > 8:
            raise RuntimeError(f"{ci} image channels: the 8-channel conv_in operand holds value + remainder of <= 4")
        # lgd_nchw_to_nhwc8_f16: channels 0..ci-1 = fp16 value, ci..2ci-1 = its rounding remainder, rest zero
        self.conv_in = (h16(pack_conv(torch.cat([w_in, w_in, w_in.new_zeros(w_in.shape[0], 8 - 2 * ci, 3, 3)], dim=1))),
                        f32(sd["encoder.conv_in.bias"]))
        n_down = 1 + max(int(m.group(1)) for k in sd for m in [re.match(r"encoder\.down_blocks\.(\d+)\.", k)] if m)
        self.downs = []
        for i in range(n_down):
            n_res = 1 + max(int(m.group(1)) for k in sd
                            for m in [re.match(rf"encoder\.down_blocks\.{i}\.resnets\.(\d+)\.", k)] if m)
            dn = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            self.downs.append(([res(f"encoder.down_blocks.{i}.resnets.{j}") for j in range(n_res)],
                               conv(dn) if f"{dn}.weight" in sd else None))
        self.mid = [res("encoder.mid_block.resnets.0"), res("encoder.mid_block.resnets.1")]
        a = "encoder.mid_block.attentions.0"
        c_mid = sd[f"{a}.group_norm.weight"].shape[0]
        legacy = f"{a}.query.weight" in sd
        nq, nk, nv, no = ("query", "key", "value", "proj_attn") if legacy else ("to_q", "to_k", "to_v", "to_out.0")
        s4 = float(c_mid) ** -0.25
        wq, wk = sd[f"{a}.{nq}.weight"].reshape(c_mid, -1) * s4, sd[f"{a}.{nk}.weight"].reshape(c_mid, -1) * s4
        self.attn = dict(n=norm(f"{a}.group_norm"),
                         qk=(h16(torch.cat([wq, wk])), f32(torch.cat([sd[f"{a}.{nq}.bias"], sd[f"{a}.{nk}.bias"]]) * s4)),
                         v=h16(sd[f"{a}.{nv}.weight"].reshape(c_mid, -1)), vb=f32(sd[f"{a}.{nv}.bias"]),
                         o=(h16(sd[f"{a}.{no}.weight"].reshape(c_mid, -1)), f32(sd[f"{a}.{no}.bias"])))
        self.norm_out = norm("encoder.conv_norm_out")
        # conv_out (C -> 2z) followed by quant_conv (1x1, 2z -> 2z): composed exactly into one 3x3 convolution
        w_o, b_o = sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"]
        if "quant_conv.weight" in sd:
            w_o, b_o = fold_pointwise_after(w_o, b_o, sd["quant_conv.weight"], sd["quant_conv.bias"])
        self.n_moments = w_o.shape[0]
        self.conv_out = (h16(pack_conv(w_o)), f32(b_o))

    @torch.no_grad()
    def encode_moments(self, image):
        """image [B, 3, H, W] fp32 in [-1, 1] -> (mean, logvar) fp32 [B, z, H/8, W/8]; logvar clamped to [-30, 20]
        (DiagonalGaussianDistribution)."""
        ops = self.ops
        x = image.to(self.dev, torch.float32).contiguous()
        B, _, H, W = x.shape
        if H != W:
            raise RuntimeError("square images only (the refiner pass resizes to 1024 x 1024, sdxl_refinement.py:26)")
        c0 = self.conv_in[0].shape[0]
        h = torch.empty((B * H * H, c0), device=self.dev, dtype=torch.float16)
        ops.gemm_launch(ops.gemm_desc(ops.nchw_to_nhwc8(x), self.conv_in[0], h, B * H * H, c0, 72, c0=8, lda0=8, taps=9,
                                      hin=H, win=H, hout=H, wout=H, bias=self.conv_in[1], ldc=c0, splits=1))
        for blk, down in self.downs:
            for r in blk:
                h = self._res(r, h, B, H)
            if down is not None:
                full = ops.conv3x3(h, down[0], B, H, H, bias=down[1])
                C = full.shape[1]
                h = full.view(B, H, H, C)[:, 1::2, 1::2].reshape(B * (H // 2) * (H // 2), C).contiguous()
                H //= 2
        h = self._res(self.mid[0], h, B, H)
        h = self._attn(self.attn, h, B, H)
        h = self._res(self.mid[1], h, B, H)
        h = ops.groupnorm(h, B, H * H, self.groups, self.eps, self.norm_out[0], self.norm_out[1], True)
        m = ops.conv3x3(h, self.conv_out[0], B, H, H, bias=self.conv_out[1])          # [B*H*H, 2z] fp16
        m = m.view(B, H, H, self.n_moments).permute(0, 3, 1, 2).float()
        mean, logvar = m.chunk(2, dim=1)
        return mean.contiguous(), logvar.clamp(-30.0, 20.0).contiguous()

    def decode(self, z):
        raise RuntimeError("HipVAEEncoder encodes; use HipVAEDecoder for decode")
