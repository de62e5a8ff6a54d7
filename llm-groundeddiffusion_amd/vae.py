"""VAE decoder ([ext] diffusers AutoencoderKL, SD 1.x config) — OUT of the HIP scope (SURVEY.md §8a
P7, §8f rank 1): it stays plain PyTorch (MIOpen / rocBLAS kernels) but sits inside the images/s
window because pipelines.decode (models/pipelines.py:117-127) runs once per box and once per image.

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
