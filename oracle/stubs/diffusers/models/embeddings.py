"""Timesteps / TimestepEmbedding as in diffusers 0.18.0 (call sites unet_2d_condition.py:305,
312-318, 801-808).  The other names exist only because the reference imports them."""
import math

import torch
from torch import nn


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1,
                           scale=1, max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32,
                                                    device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels,
                                      flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None,
                 cond_proj_dim=None):
        super().__init__()
        assert act_fn == "silu" and post_act_fn is None and cond_proj_dim is None
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


def _unused(name):
    class _U(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} is not used by SD 1.x / 2.x")

    _U.__name__ = name
    return _U


GaussianFourierProjection = _unused("GaussianFourierProjection")
TextImageProjection = _unused("TextImageProjection")
TextImageTimeEmbedding = _unused("TextImageTimeEmbedding")
TextTimeEmbedding = _unused("TextTimeEmbedding")
CombinedTimestepLabelEmbeddings = _unused("CombinedTimestepLabelEmbeddings")
ImagePositionalEmbeddings = _unused("ImagePositionalEmbeddings")
PatchEmbed = _unused("PatchEmbed")
