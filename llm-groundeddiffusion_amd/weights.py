"""UNet configurations, parameter inventory (diffusers / reference naming) and the seeded synthetic
weight factory.

There are no checkpoints in the sandbox (no network), so benchmarks and parity tests use random
weights of the exact SD architectures, generated *by parameter name* so that the CPU oracle, the
reference harness and the HIP engine all see bit-identical fp32 tensors on any machine:

    value(name) = f(torch.Generator().manual_seed(crc32(name) ^ seed), shape, kind(name))

Naming follows the reference's modules (models/unet_2d_condition.py:289-572, unet_2d_blocks.py,
transformer_2d.py:140-214, attention.py:25-154, attention_processor.py:127-143) so real
`UNet2DConditionModel.state_dict()` checkpoints load through the same path.
"""
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch


@dataclass(frozen=True)
class UNetConfig:
    """Subset of the reference constructor (unet_2d_condition.py:201-251) that SD 1.x / 2.x use."""
    name: str = "sd15"
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    attention_head_dim: Tuple[int, ...] = (8, 8, 8, 8)  # = number of heads per level (sic)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False
    use_gated_attention: bool = False   # GLIGEN fuser + position_net
    sample_size: int = 64
    prediction_type: str = "epsilon"
    gligen_positive_len: int = 768      # hard-coded at unet_2d_condition.py:572
    # --- SDXL-refiner topology (generation/sdxl_refinement.py:13-15; [ext] diffusers' SDXL UNet2DConditionModel) ---
    down_attn: Optional[Tuple[bool, ...]] = None   # per down block: CrossAttnDownBlock2D?  None = all but the last (SD)
    up_attn: Optional[Tuple[bool, ...]] = None     # per up block: CrossAttnUpBlock2D?      None = all but the first (SD)
    transformer_depth: int = 1                     # BasicTransformerBlocks per Transformer2DModel (transformer_layers_per_block)
    addition_embed_type: Optional[str] = None      # "text_time": pooled text embedding + size / crop / score ids
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 0  # pooled width + 5 * addition_time_embed_dim for the refiner

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4

    @property
    def pooled_dim(self):
        """Width of `text_embeds` (the projected pooled output of the text tower) of a text_time model."""
        return self.projection_class_embeddings_input_dim - 5 * self.addition_time_embed_dim

    def to_ref_kwargs(self):
        """kwargs for the reference's UNet2DConditionModel(...) (oracle harness)."""
        if self.transformer_depth != 1 or self.addition_embed_type is not None:
            raise ValueError("the reference's UNet (diffusers 0.18 lineage) has neither multi-layer transformer blocks "
                             "nor text_time conditioning")
        extra = {}
        if self.down_attn is not None:
            extra["down_block_types"] = tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in self.down_attn)
        if self.up_attn is not None:
            extra["up_block_types"] = tuple("CrossAttnUpBlock2D" if a else "UpBlock2D" for a in self.up_attn)
        return dict(**extra, sample_size=self.sample_size, in_channels=self.in_channels,
                    out_channels=self.out_channels, block_out_channels=self.block_out_channels,
                    layers_per_block=self.layers_per_block,
                    cross_attention_dim=self.cross_attention_dim,
                    attention_head_dim=self.attention_head_dim,
                    norm_num_groups=self.norm_num_groups, norm_eps=self.norm_eps,
                    use_linear_projection=self.use_linear_projection,
                    use_gated_attention=self.use_gated_attention)


CONFIGS = {
    # tiny configs: same topology, channels multiples of 64 so every kernel constraint is exercised
    "tiny": UNetConfig(name="tiny", block_out_channels=(64, 128, 256, 256), cross_attention_dim=128,
                       sample_size=32),
    "tiny_gligen": UNetConfig(name="tiny_gligen", block_out_channels=(64, 128, 256, 256),
                              cross_attention_dim=768, use_gated_attention=True, sample_size=32),
    # SD2.x-style topology in small: linear proj_in/proj_out, heads per level chosen so that every head is
    # 64 wide (the SD2.1 head dim), text width != 768
    "tiny_sd21": UNetConfig(name="tiny_sd21", block_out_channels=(64, 128, 256, 256), cross_attention_dim=192,
                            attention_head_dim=(1, 2, 4, 4), use_linear_projection=True, sample_size=32),
    "sd15": UNetConfig(name="sd15"),
    "sd14_gligen": UNetConfig(name="sd14_gligen", use_gated_attention=True),
    # SDXL refiner 1.0 ([ext] stabilityai/stable-diffusion-xl-refiner-1.0 unet/config.json): outer blocks without
    # attention, four transformer layers per attention block, every head 64 wide, OpenCLIP-bigG text width 1280,
    # text_time conditioning over the 1280-wide pooled embedding + (orig h, w, crop top, left, aesthetic score)
    "sdxl_refiner": UNetConfig(name="sdxl_refiner", block_out_channels=(384, 768, 1536, 1536), cross_attention_dim=1280,
                               attention_head_dim=(6, 12, 24, 24), use_linear_projection=True, sample_size=128,
                               down_attn=(False, True, True, False), up_attn=(False, True, True, False),
                               transformer_depth=4, addition_embed_type="text_time", addition_time_embed_dim=256,
                               projection_class_embeddings_input_dim=2560),
    # the same topology in small: 64-wide heads, depth 2, pooled width 96 + 5 x 32 ids
    "tiny_xl": UNetConfig(name="tiny_xl", block_out_channels=(64, 128, 256, 256), cross_attention_dim=128,
                          attention_head_dim=(1, 2, 4, 4), use_linear_projection=True, sample_size=32,
                          down_attn=(False, True, True, False), up_attn=(False, True, True, False),
                          transformer_depth=2, addition_embed_type="text_time", addition_time_embed_dim=32,
                          projection_class_embeddings_input_dim=256),
    # topology-only variant the REFERENCE's own UNet class can build (depth 1, no added conditioning): pins the
    # outer-blocks-without-attention wiring against /root/reference (tests/golden/unet_fwd_tiny_outer.npz)
    "tiny_outer": UNetConfig(name="tiny_outer", block_out_channels=(64, 128, 256, 256), cross_attention_dim=128,
                             attention_head_dim=(1, 2, 4, 4), use_linear_projection=True, sample_size=32,
                             down_attn=(False, True, True, False), up_attn=(False, True, True, False)),
    "sd21": UNetConfig(name="sd21", cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20),
                       use_linear_projection=True, sample_size=96, prediction_type="v_prediction"),
}


# -------------------------------------------------------------------------------------------------
# structure
# -------------------------------------------------------------------------------------------------
@dataclass
class ResnetSpec:
    prefix: str
    cin: int          # total input channels (hidden + skip)
    cskip: int        # channels that come from the skip connection (0 for down/mid)
    cout: int

    @property
    def shortcut(self):
        return self.cin != self.cout


@dataclass
class AttnSpec:
    prefix: str       # "...attentions.j"
    key: Tuple        # ("down", i, j, 0) — the attn_key of pipelines.py:12
    channels: int
    heads: int
    depth: int = 1    # transformer_blocks.0 .. depth-1; the attn key of layer d is key[:3] + (d,)

    @property
    def head_dim(self):
        return self.channels // self.heads


@dataclass
class BlockSpec:
    kind: str                      # "down" | "mid" | "up"
    index: int
    resnets: List[ResnetSpec] = field(default_factory=list)
    attns: List[AttnSpec] = field(default_factory=list)
    sampler: Optional[str] = None  # prefix of the down/upsampler conv, if any
    channels: int = 0
    level: int = 0                 # resolution level: 0 = full latent size, 1 = /2, ...


def unet_blocks(cfg: UNetConfig) -> List[BlockSpec]:
    """Block wiring exactly as unet_2d_condition.py:440-543 builds it."""
    boc = cfg.block_out_channels
    n = len(boc)
    blocks = []
    out_c = boc[0]
    for i in range(n):
        in_c, out_c = out_c, boc[i]
        # down_block_types default: 3x CrossAttnDownBlock2D + DownBlock2D
        has_attn = cfg.down_attn[i] if cfg.down_attn is not None else i < n - 1
        b = BlockSpec("down", i, channels=out_c, level=i)
        for j in range(cfg.layers_per_block):
            b.resnets.append(ResnetSpec(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, 0, out_c))
            if has_attn:
                b.attns.append(AttnSpec(f"down_blocks.{i}.attentions.{j}", ("down", i, j, 0), out_c,
                                        cfg.attention_head_dim[i], cfg.transformer_depth))
        if i < n - 1:
            b.sampler = f"down_blocks.{i}.downsamplers.0.conv"
        blocks.append(b)
    mid = BlockSpec("mid", 0, channels=boc[-1], level=n - 1)
    mid.resnets = [ResnetSpec("mid_block.resnets.0", boc[-1], 0, boc[-1]),
                   ResnetSpec("mid_block.resnets.1", boc[-1], 0, boc[-1])]
    mid.attns = [AttnSpec("mid_block.attentions.0", ("mid", 0, 0, 0), boc[-1], cfg.attention_head_dim[-1],
                          cfg.transformer_depth)]
    blocks.append(mid)
    rev = list(reversed(boc))
    rev_heads = list(reversed(cfg.attention_head_dim))
    out_c = rev[0]
    for i in range(n):
        prev_out = out_c
        out_c = rev[i]
        in_c = rev[min(i + 1, n - 1)]
        # up_block_types default: UpBlock2D + 3x CrossAttnUpBlock2D
        has_attn = cfg.up_attn[i] if cfg.up_attn is not None else i > 0
        b = BlockSpec("up", i, channels=out_c, level=n - 1 - i)
        nl = cfg.layers_per_block + 1
        for j in range(nl):
            skip = in_c if j == nl - 1 else out_c
            hid = prev_out if j == 0 else out_c
            b.resnets.append(ResnetSpec(f"up_blocks.{i}.resnets.{j}", hid + skip, skip, out_c))
            if has_attn:
                b.attns.append(AttnSpec(f"up_blocks.{i}.attentions.{j}", ("up", i, j, 0), out_c, rev_heads[i],
                                        cfg.transformer_depth))
        if i < n - 1:
            b.sampler = f"up_blocks.{i}.upsamplers.0.conv"
        blocks.append(b)
    return blocks


ALL_ATTN_KEYS = None  # filled lazily


def attn_keys(cfg: UNetConfig):
    return [a.key for b in unet_blocks(cfg) for a in b.attns]


# -------------------------------------------------------------------------------------------------
# parameter inventory
# -------------------------------------------------------------------------------------------------
def _attention_params(p, prefix, query_dim, cross_dim, inner):
    p[f"{prefix}.to_q.weight"] = (inner, query_dim)
    p[f"{prefix}.to_k.weight"] = (inner, cross_dim)
    p[f"{prefix}.to_v.weight"] = (inner, cross_dim)
    p[f"{prefix}.to_out.0.weight"] = (query_dim, inner)
    p[f"{prefix}.to_out.0.bias"] = (query_dim,)


def _ff_params(p, prefix, dim):
    p[f"{prefix}.net.0.proj.weight"] = (8 * dim, dim)
    p[f"{prefix}.net.0.proj.bias"] = (8 * dim,)
    p[f"{prefix}.net.2.weight"] = (dim, 4 * dim)
    p[f"{prefix}.net.2.bias"] = (dim,)


def _norm(p, prefix, c):
    p[f"{prefix}.weight"] = (c,)
    p[f"{prefix}.bias"] = (c,)


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    p: Dict[str, Tuple[int, ...]] = {}
    c0 = cfg.block_out_channels[0]
    ted = cfg.time_embed_dim
    p["conv_in.weight"] = (c0, cfg.in_channels, 3, 3)
    p["conv_in.bias"] = (c0,)
    p["time_embedding.linear_1.weight"] = (ted, c0)
    p["time_embedding.linear_1.bias"] = (ted,)
    p["time_embedding.linear_2.weight"] = (ted, ted)
    p["time_embedding.linear_2.bias"] = (ted,)
    if cfg.addition_embed_type == "text_time":          # [ext] TimestepEmbedding(projection_class_embeddings_input_dim, ted)
        p["add_embedding.linear_1.weight"] = (ted, cfg.projection_class_embeddings_input_dim)
        p["add_embedding.linear_1.bias"] = (ted,)
        p["add_embedding.linear_2.weight"] = (ted, ted)
        p["add_embedding.linear_2.bias"] = (ted,)
    for b in unet_blocks(cfg):
        for r in b.resnets:
            _norm(p, f"{r.prefix}.norm1", r.cin)
            p[f"{r.prefix}.conv1.weight"] = (r.cout, r.cin, 3, 3)
            p[f"{r.prefix}.conv1.bias"] = (r.cout,)
            p[f"{r.prefix}.time_emb_proj.weight"] = (r.cout, ted)
            p[f"{r.prefix}.time_emb_proj.bias"] = (r.cout,)
            _norm(p, f"{r.prefix}.norm2", r.cout)
            p[f"{r.prefix}.conv2.weight"] = (r.cout, r.cout, 3, 3)
            p[f"{r.prefix}.conv2.bias"] = (r.cout,)
            if r.shortcut:
                p[f"{r.prefix}.conv_shortcut.weight"] = (r.cout, r.cin, 1, 1)
                p[f"{r.prefix}.conv_shortcut.bias"] = (r.cout,)
        for a in b.attns:
            C = a.channels
            _norm(p, f"{a.prefix}.norm", C)
            proj_shape = (C, C) if cfg.use_linear_projection else (C, C, 1, 1)
            p[f"{a.prefix}.proj_in.weight"] = proj_shape
            p[f"{a.prefix}.proj_in.bias"] = (C,)
            for d in range(a.depth):
                t = f"{a.prefix}.transformer_blocks.{d}"
                _norm(p, f"{t}.norm1", C)
                _attention_params(p, f"{t}.attn1", C, C, C)
                _norm(p, f"{t}.norm2", C)
                _attention_params(p, f"{t}.attn2", C, cfg.cross_attention_dim, C)
                _norm(p, f"{t}.norm3", C)
                _ff_params(p, f"{t}.ff", C)
                if cfg.use_gated_attention:
                    f = f"{t}.fuser"
                    p[f"{f}.linear.weight"] = (C, cfg.cross_attention_dim)
                    p[f"{f}.linear.bias"] = (C,)
                    _attention_params(p, f"{f}.attn", C, C, C)
                    _ff_params(p, f"{f}.ff", C)
                    _norm(p, f"{f}.norm1", C)
                    _norm(p, f"{f}.norm2", C)
                    p[f"{f}.alpha_attn"] = ()
                    p[f"{f}.alpha_dense"] = ()
            p[f"{a.prefix}.proj_out.weight"] = proj_shape
            p[f"{a.prefix}.proj_out.bias"] = (C,)
        if b.sampler:
            p[f"{b.sampler}.weight"] = (b.channels, b.channels, 3, 3)
            p[f"{b.sampler}.bias"] = (b.channels,)
    _norm(p, "conv_norm_out", c0)
    p["conv_out.weight"] = (cfg.out_channels, c0, 3, 3)
    p["conv_out.bias"] = (cfg.out_channels,)
    if cfg.use_gated_attention:
        pl, pd = cfg.gligen_positive_len, 64
        p["position_net.linears.0.weight"] = (512, pl + pd)
        p["position_net.linears.0.bias"] = (512,)
        p["position_net.linears.2.weight"] = (512, 512)
        p["position_net.linears.2.bias"] = (512,)
        p["position_net.linears.4.weight"] = (cfg.cross_attention_dim, 512)
        p["position_net.linears.4.bias"] = (cfg.cross_attention_dim,)
        p["position_net.null_positive_feature"] = (pl,)
        p["position_net.null_position_feature"] = (pd,)
    return p


def num_params(cfg):
    n = 0
    for s in param_shapes(cfg).values():
        k = 1
        for d in s:
            k *= d
        n += k
    return n


# -------------------------------------------------------------------------------------------------
# synthetic weights
# -------------------------------------------------------------------------------------------------
def synth_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("alpha_attn", "alpha_dense"):
        return torch.tensor(1.0)                       # GLIGEN gates open (they initialise to 0)
    if leaf.startswith("null_"):
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if len(shape) == 1:                                # norm scale
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    w = (torch.rand(shape, generator=g) * 2 - 1) * (fan_in ** -0.5)   # U(-1/sqrt(fan_in), +)
    if ".attn2.to_q." in name or ".attn2.to_k." in name:
        w = w * 4.0                                    # O(1) logit spread -> non-degenerate maps
    return w


def synth_state_dict(cfg: UNetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {n: synth_tensor(n, s, seed) for n, s in param_shapes(cfg).items()}


def synth_embeddings(cfg: UNetConfig, n_cond: int = 1, seed: int = 1):
    """(uncond (1,77,Cx), cond (n,77,Cx)) stand-ins for CLIP hidden states ("cached layouts")."""
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    cond = torch.randn((n_cond, 77, cfg.cross_attention_dim), generator=g)
    g2 = torch.Generator(device="cpu").manual_seed(2000 + seed)
    uncond = torch.randn((1, 77, cfg.cross_attention_dim), generator=g2)
    return uncond, cond
