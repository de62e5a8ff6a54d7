"""VAE decoder ([ext] diffusers AutoencoderKL, SD 1.x config).  It sits inside the images/s window
because pipelines.decode (models/pipelines.py:117-127) runs once per box and once per image
(SURVEY.md §8a P7, §8f rank 1).

Two implementations of the same module:
  * VAEDecoder      — plain PyTorch (reference for tests; on a fresh MI355X box MIOpen falls back to
                      its naive convolution, 0.16 s per conv, which would dominate the benchmark);
  * HipVAEDecoder   — the same weights run through the engine's own kernels (implicit-GEMM conv,
                      GroupNorm+SiLU, GEMM + row softmax for the single 512-channel attention).

Random-init weights of the exact decoder architecture (there are no checkpoints in the sandbox):
post_quant_conv 4->4, conv_in 4->512, mid (resnet, 1-head attention, resnet), 4 up blocks
(512,512,256,128) x 3 resnets with 3 nearest-2x upsamplers, GroupNorm+SiLU, conv_out 128->3.
"""
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


def make_vae(device, dtype=torch.float16, seed=0):
    g = torch.random.get_rng_state()
    torch.manual_seed(1234 + seed)
    vae = VAEDecoder()
    torch.random.set_rng_state(g)
    return vae.to(device=device, dtype=dtype).eval()


class HipVAEDecoder:
    """VAEDecoder.decode on the lgd_hip kernels (channels-last fp16, fp32 accumulate)."""

    def __init__(self, vae: VAEDecoder, device):
        from . import ops
        from .weightstore import pack_conv
        self.ops = ops
        self.dev = torch.device(device)
        h16 = lambda t: t.detach().to(self.dev, torch.float16).contiguous()
        f32 = lambda t: t.detach().to(self.dev, torch.float32).contiguous()
        conv = lambda m: (h16(pack_conv(m.weight.detach().float())), f32(m.bias))
        lin = lambda m: (h16(m.weight.detach().float().reshape(m.weight.shape[0], -1)), f32(m.bias))
        norm = lambda m: (f32(m.weight), f32(m.bias))
        res = lambda r: dict(n1=norm(r.norm1), c1=conv(r.conv1), n2=norm(r.norm2), c2=conv(r.conv2),
                             sc=lin(r.short) if r.short is not None else None)
        self.pq_w = f32(vae.post_quant_conv.weight.reshape(vae.post_quant_conv.weight.shape[0], -1))
        self.pq_b = f32(vae.post_quant_conv.bias)
        self.conv_in = conv(vae.conv_in)
        a = vae.mid[1]
        self.mid = [res(vae.mid[0]), dict(n=norm(a.norm), q=lin(a.q), k=lin(a.k), v=lin(a.v), o=lin(a.o)),
                    res(vae.mid[2])]
        self.ups = [([res(r) for r in blk], conv(up) if up is not None else None) for blk, up in vae.ups]
        self.norm_out = norm(vae.norm_out)
        w = pack_conv(vae.conv_out.weight.detach().float())              # [3, 9*C]
        self.conv_out = (h16(torch.cat([w, torch.zeros(1, w.shape[1])])),    # pad to 4 output channels
                         f32(torch.cat([vae.conv_out.bias.detach().float().cpu(), torch.zeros(1)])))

    def _res(self, p, x, B, H):
        ops = self.ops
        HW = H * H
        h = ops.groupnorm(x, B, HW, 32, 1e-6, p["n1"][0], p["n1"][1], True)
        h = ops.conv3x3(h, p["c1"][0], B, H, H, bias=p["c1"][1])
        h = ops.groupnorm(h, B, HW, 32, 1e-6, p["n2"][0], p["n2"][1], True)
        sc = x if p["sc"] is None else ops.linear(x, p["sc"][0], p["sc"][1])
        return ops.conv3x3(h, p["c2"][0], B, H, H, bias=p["c2"][1], res=sc)

    def _attn(self, p, x, B, H):
        ops = self.ops
        S = H * H
        C = x.shape[1]
        n = ops.groupnorm(x, B, S, 32, 1e-6, p["n"][0], p["n"][1], False)
        q, k, v = (ops.linear(n, p[t][0], p[t][1]) for t in "qkv")
        outs = []
        for b in range(B):
            sl = slice(b * S, (b + 1) * S)
            sc = ops.linear(q[sl], k[sl])                                 # [S, S] = q k^T
            pr = ops.softmax_rows(sc, C ** -0.5)
            outs.append(ops.linear(pr, v[sl].t().contiguous()))           # [S, C]
        a = outs[0] if B == 1 else torch.cat(outs)
        return ops.linear(a, p["o"][0], p["o"][1], res=x)

    @torch.no_grad()
    def decode(self, z):
        ops = self.ops
        z = z.to(self.dev, torch.float32)
        B, _, L, _ = z.shape
        z = torch.einsum("oc,bchw->bohw", self.pq_w, z) + self.pq_b.view(1, -1, 1, 1)
        h = ops.conv_in(z.contiguous(), self.conv_in[0], self.conv_in[1])
        H = L
        h = self._res(self.mid[0], h, B, H)
        h = self._attn(self.mid[1], h, B, H)
        h = self._res(self.mid[2], h, B, H)
        for blk, up in self.ups:
            for r in blk:
                h = self._res(r, h, B, H)
            if up is not None:
                h = ops.conv3x3(h, up[0], B, H, H, bias=up[1], ups=1)
                H *= 2
        h = ops.groupnorm(h, B, H * H, 32, 1e-6, self.norm_out[0], self.norm_out[1], True)
        y = ops.conv_out(h, self.conv_out[0], self.conv_out[1], B, H)
        return y[:, :3]


def make_hip_vae(device, seed=0):
    """Random-init SD VAE decoder running on the HIP kernels."""
    g = torch.random.get_rng_state()
    torch.manual_seed(1234 + seed)
    vae = VAEDecoder()
    torch.random.set_rng_state(g)
    return HipVAEDecoder(vae, device)
