"""ResnetBlock2D / Downsample2D / Upsample2D as in diffusers 0.18.0 (restated; constructed by
keyword at unet_2d_blocks.py:186-197, 315-326, 360-362, 477-488, 577-588, 621, 738-749)."""
import torch.nn.functional as F
from torch import nn


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None,
                 name="conv"):
        super().__init__()
        assert use_conv and not use_conv_transpose
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None):
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        return self.conv(hidden_states)


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv and padding == 1
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states):
        return self.conv(hidden_states)


class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0,
                 temb_channels=512, groups=32, groups_out=None, pre_norm=True, eps=1e-6,
                 non_linearity="swish", time_embedding_norm="default", kernel=None,
                 output_scale_factor=1.0, use_in_shortcut=None, up=False, down=False,
                 conv_shortcut_bias=True, conv_2d_out_channels=None, skip_time_act=False):
        super().__init__()
        assert time_embedding_norm == "default" and not up and not down
        assert non_linearity in ("swish", "silu")
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.output_scale_factor = output_scale_factor
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.use_in_shortcut = in_channels != out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = None
        if self.use_in_shortcut:
            self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1,
                                           padding=0, bias=conv_shortcut_bias)

    def forward(self, input_tensor, temb):
        hidden_states = self.nonlinearity(self.norm1(input_tensor))
        hidden_states = self.conv1(hidden_states)
        if self.time_emb_proj is not None:
            temb = self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
            hidden_states = hidden_states + temb
        hidden_states = self.nonlinearity(self.norm2(hidden_states))
        hidden_states = self.dropout(hidden_states)
        hidden_states = self.conv2(hidden_states)
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + hidden_states) / self.output_scale_factor
