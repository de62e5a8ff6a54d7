"""ORACLE TEST INFRASTRUCTURE — plain PyTorch fp32 restatement of the AutoencoderKL DECODER ([ext] diffusers 0.18.0, SD 1.x /
2.x config: post_quant_conv, conv_in, mid block with one single-head attention, four up blocks of three resnets, GroupNorm +
SiLU, conv_out).  The parity reference of the VAE tests (tests/test_engine_gpu.py, test_dropin_gpu.py, test_boxdiff_gpu.py):
HipVAEDecoder runs `VAEDecoder.aekl_state_dict()` (AutoencoderKL's key names) on the HIP kernels and is compared with
`VAEDecoder.decode` at the full SD size.  Moved out of the product package in round 6 (a test oracle must not live
inside it); the package keeps only the seeded weight factory (`lgd_amd.vae.synth_aekl_state_dict`).

Parity note: AutoencoderKL itself is absent from the sandbox, so this is a restatement from the published architecture —
"parity unpinned" at that boundary.  Only tests/, smoke() and bench.py's cpu_baseline leg may import oracle/."""
import torch
import torch.nn.functional as F
from torch import nn


class _Res(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.short = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.short is None else self.short(x)) + h


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = nn.GroupNorm(32, c, eps=1e-6)
        self.q, self.k, self.v, self.o = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.norm(x).reshape(B, C, H * W).transpose(1, 2)
        q, k, v = self.q(h), self.k(h), self.v(h)
        a = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        return x + self.o(a).transpose(1, 2).reshape(B, C, H, W)


class VAEDecoder(nn.Module):
    def __init__(self, latent_channels=4, ch=(512, 512, 256, 128), layers=3):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.conv_in = nn.Conv2d(latent_channels, ch[0], 3, padding=1)
        self.mid = nn.Sequential(_Res(ch[0], ch[0]), _Attn(ch[0]), _Res(ch[0], ch[0]))
        ups = []
        cin = ch[0]
        for i, c in enumerate(ch):
            blk = [_Res(cin if j == 0 else c, c) for j in range(layers)]
            cin = c
            ups.append(nn.ModuleList([nn.Sequential(*blk), nn.Conv2d(c, c, 3, padding=1) if i < len(ch) - 1 else None]))
        self.ups = nn.ModuleList(ups)
        self.norm_out = nn.GroupNorm(32, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 3, 3, padding=1)

    @torch.no_grad()
    def decode(self, z):
        z = z.to(self.conv_in.weight.dtype)
        h = self.conv_in(self.post_quant_conv(z))
        h = self.mid(h)
        for blk, up in self.ups:
            h = blk(h)
            if up is not None:
                h = up(F.interpolate(h, scale_factor=2.0, mode="nearest"))
        return self.conv_out(F.silu(self.norm_out(h)))

    def aekl_state_dict(self, legacy_attention_names=False):
        """The parameters under AutoencoderKL's names (diffusers: `post_quant_conv.*`, `decoder.conv_in.*`,
        `decoder.mid_block.resnets.N.*`, `decoder.mid_block.attentions.0.{group_norm,to_q,to_k,to_v,to_out.0}.*`,
        `decoder.up_blocks.I.resnets.J.{norm1,conv1,norm2,conv2,conv_shortcut}.*`, `decoder.up_blocks.I.upsamplers.0.conv.*`,
        `decoder.conv_norm_out.*`, `decoder.conv_out.*`).  legacy_attention_names: the pre-0.15 AttentionBlock names
        (`query / key / value / proj_attn`) that SD checkpoints on the hub still carry."""
        out = {}

        def put(dst, m):
            out[f"{dst}.weight"] = m.weight.detach().clone()
            out[f"{dst}.bias"] = m.bias.detach().clone()

        def res(dst, r):
            put(f"{dst}.norm1", r.norm1); put(f"{dst}.conv1", r.conv1)
            put(f"{dst}.norm2", r.norm2); put(f"{dst}.conv2", r.conv2)
            if r.short is not None:
                put(f"{dst}.conv_shortcut", r.short)
        put("post_quant_conv", self.post_quant_conv)
        put("decoder.conv_in", self.conv_in)
        res("decoder.mid_block.resnets.0", self.mid[0])
        res("decoder.mid_block.resnets.1", self.mid[2])
        a = "decoder.mid_block.attentions.0"
        names = dict(norm="group_norm", q="query", k="key", v="value", o="proj_attn") if legacy_attention_names else \
            dict(norm="group_norm", q="to_q", k="to_k", v="to_v", o="to_out.0")
        for k, n in names.items():
            put(f"{a}.{n}", getattr(self.mid[1], k))
        for i, (blk, up) in enumerate(self.ups):
            for j, r in enumerate(blk):
                res(f"decoder.up_blocks.{i}.resnets.{j}", r)
            if up is not None:
                put(f"decoder.up_blocks.{i}.upsamplers.0.conv", up)
        put("decoder.conv_norm_out", self.norm_out)
        put("decoder.conv_out", self.conv_out)
        return out
