"""TEST INFRASTRUCTURE — a torch restatement of the few `lgd_amd.ops` entry points that `lgd_amd/sam.py` calls, with the
same argument conventions (flattened channels-last tokens, (ld, batch-stride) views, fp16 storage, fp32 arithmetic).

It exists so that the HOST algebra of the SAM port (weight re-packing, the rel-pos-bias-as-extra-head-columns trick,
transposed convolutions as GEMMs, the nested pixel order of the mask head, pre-projected position tables) can be
checked against the Hugging Face module on the GPU-less build container (tests/test_sam_host.py), and so that the HIP
kernels of csrc/sam.hip have an independent statement to be compared with on the GPU (tests/test_sam_gpu.py).
The product never imports this file: `lgd_amd.ops` raises without liblgd_hip.so.
"""
import torch
import torch.nn.functional as F

F16, F32 = torch.float16, torch.float32
ACT_GELU, ACT_RELU = 1, 2


def linear(x, w, bias=None, res=None, out=None, *, out_f32=False, **_kw):
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    if res is not None:
        y = y + res.float()
    y = y.to(F32 if out_f32 else F16)
    if out is not None:
        out.copy_(y)
        return out
    return y


def layernorm(x, gamma, beta, eps=1e-5, **_kw):
    C = gamma.shape[0]
    return F.layer_norm(x.float().reshape(-1, C), (C,), gamma.float(), beta.float(), eps).to(F16)


def conv3x3(x, w, B, H, W, **_kw):
    Cin, Cout = x.shape[1], w.shape[0]
    wt = w.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(x.float().reshape(B, H, W, Cin).permute(0, 3, 1, 2), wt, padding=1)
    return y.permute(0, 2, 3, 1).reshape(B * H * W, Cout).to(F16)


def act(x, mode, out=None):
    return (F.gelu(x.float()) if mode == ACT_GELU else F.relu(x.float())).to(F16)


def add(a, b, out=None):
    return (a.float() + b.float()).to(F16)


def _heads(t, view, B, S, H, d):
    ld, bs = view
    return torch.as_strided(t, (B, S, H, d), (bs, ld, d, 1)).float().permute(0, 2, 1, 3)      # B, H, S, d


def attn_fwd(q, k, v, o, B, H, Sq, Sk, d, scale, *, lse=None, q_view=None, k_view=None, v_view=None, o_view=None):
    qh = _heads(q, q_view or (H * d, Sq * H * d), B, Sq, H, d)
    kh = _heads(k, k_view or (H * d, Sk * H * d), B, Sk, H, d)
    vh = _heads(v, v_view or (H * d, Sk * H * d), B, Sk, H, d)
    p = torch.softmax(scale * qh @ kh.transpose(-1, -2), dim=-1)
    out = (p @ vh).permute(0, 2, 1, 3).reshape(B * Sq, H * d)
    assert o_view is None
    o.copy_(out.to(F16))
    return o


def sam_relpos_qkv(qkv, qkv_bias, rel_h, rel_w, B, Hs, Ws, window, NH, d, DA, scale):
    """Statement of lgd_sam_relpos_qkv_f16 (include/lgd_hip.h)."""
    S = window or Hs
    C = NH * d
    nwy, nwx = -(-Hs // S), -(-Ws // S)
    grid = qkv_bias.float().to(F16).float().reshape(1, 1, 1, 3 * C).repeat(B, nwy * S, nwx * S, 1)   # padding = projection of 0
    grid[:, :Hs, :Ws] = qkv.float().reshape(B, Hs, Ws, 3 * C)
    win = grid.reshape(B, nwy, S, nwx, S, 3, NH, d).permute(0, 1, 3, 2, 4, 5, 6, 7)                  # b wy wx iy ix 3 NH d
    q, k, v = win[..., 0, :, :], win[..., 1, :, :], win[..., 2, :, :]
    dv = qkv.device
    idx = torch.arange(S, device=dv)
    rel = idx[:, None] - idx[None, :] + S - 1                                                         # [pos, j]
    Rh, Rw = rel_h.float()[rel], rel_w.float()[rel]                                                   # [S, S, d]
    bias_h = torch.einsum("bwvyxhc,yjc->bwvyxhj", q, Rh) / scale
    bias_w = torch.einsum("bwvyxhc,xjc->bwvyxhj", q, Rw) / scale
    eye = torch.eye(S, device=dv)
    oh_y = eye[idx][None, None, None, :, None, None, :].expand(B, nwy, nwx, S, S, NH, S)
    oh_x = eye[idx][None, None, None, None, :, None, :].expand(B, nwy, nwx, S, S, NH, S)
    pad = torch.zeros(B, nwy, nwx, S, S, NH, DA - d - 2 * S, device=dv)
    z = torch.zeros(B, nwy, nwx, S, S, NH, DA - d, device=dv)
    rows = B * nwy * nwx * S * S
    pack = lambda parts: torch.cat(parts, dim=-1).reshape(rows, NH * DA).to(F16)
    return pack([q, bias_h, bias_w, pad]), pack([k, oh_y, oh_x, pad]), pack([v, z])


def sam_window_merge(oa, B, Hs, Ws, window, NH, d, DA):
    S = window or Hs
    nwy, nwx = -(-Hs // S), -(-Ws // S)
    t = oa.reshape(B, nwy, nwx, S, S, NH, DA)[..., :d].permute(0, 1, 3, 2, 4, 5, 6).reshape(B, nwy * S, nwx * S, NH * d)
    return t[:, :Hs, :Ws].reshape(B * Hs * Ws, NH * d).contiguous()
