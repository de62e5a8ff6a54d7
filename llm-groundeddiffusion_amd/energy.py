"""Host side of the cross-attention energy (utils/guidance.py:91-286): turns the layout (boxes, token
positions, hyper-parameters) into the item / coefficient / mask tables that the single-launch HIP
kernel `lgd_ca_energy_f32` consumes.  Everything here is tiny integer/box bookkeeping and follows
the reference's rounding rules exactly (utils/utils.py:57-70: Python round = banker's rounding).
"""
import math
from collections.abc import Iterable
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .hostprep import scale_proportion

F32 = torch.float32


def box_mask(obj_boxes, H, W) -> torch.Tensor:
    """guidance.py:104-114: union of an object's boxes at map resolution."""
    mask = torch.zeros(H, W)
    if not isinstance(obj_boxes[0], Iterable):
        obj_boxes = [obj_boxes]
    for b in obj_boxes:
        x0, y0, x1, y1 = scale_proportion(b, H, W)
        mask[y0:y1, x0:x1] = 1
    return mask


class EnergyTables:
    """Static description of one guidance problem (one layout): built once per run() call."""

    def __init__(self, device, bboxes, object_positions, guidance_attn_keys: Sequence[Tuple],
                 map_hw: Dict[Tuple, int], heads: int, text_len: int = 77, *, loss_scale=30.0,
                 use_ratio_based_loss=True, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=1.0,
                 ref_boxes: bool = False, ref_ca_loss_weight=1.0, ref_ca_word_token_only=False,
                 ref_ca_last_token_only=True, word_token_indices=None):
        # `use_ratio_based_loss` defaults to True because the reference's add_ca_loss_per_attn_map_to_loss does
        # (guidance.py:91): LMD / LMD+ switch it off explicitly (lmd_plus.py:315,487; lmd.py:349,521), the
        # backward_guidance plugin does not (backward_guidance.py:99-112) and runs the ratio branch (:118-130).
        self.device = device
        self.use_ratio_based_loss = bool(use_ratio_based_loss)
        self.keys = [tuple(k) for k in guidance_attn_keys]
        self.heads, self.T = heads, text_len
        self.n_obj = len(bboxes)
        self.max_hw = max([map_hw[k] for k in self.keys] + [1])
        self._map_hw = dict(map_hw)
        n_keys = len(self.keys)
        items, coefs, masks = [], [], []
        self.ref_slots: List[Tuple[int, int, int]] = []     # (obj, box, key index) per ref_id
        denom = max(self.n_obj * n_keys, 1)                 # guidance.py:270,284
        for ki, key in enumerate(self.keys):
            hw = map_hw[key]
            Hs = int(math.sqrt(hw))
            for o in range(self.n_obj):
                m = box_mask(bboxes[o], Hs, Hs)
                # guidance.py:136-137 — computed with the same fp32 tensor ops as the reference
                k_fg = int((m.sum() * fg_top_p).long().clamp_(min=1))
                k_bg = int(((1 - m).sum() * bg_top_p).long().clamp_(min=1))
                mid = len(masks)
                masks.append(torch.nn.functional.pad(m.reshape(-1), (0, self.max_hw - hw)))
                toks = object_positions[o]
                for p in toks:
                    if use_ratio_based_loss:                 # guidance.py:124-126: mean over heads of (1 - r)^2
                        items.append([ki, 2, int(p), mid, 1, 1, 0, 0])
                        coefs.append([0.0, 0.0, 0.0, loss_scale / (heads * len(toks) * denom)])
                        continue
                    items.append([ki, 0, int(p), mid, k_fg, k_bg, 0, 0])
                    coefs.append([loss_scale * fg_weight / (len(toks) * denom),
                                  loss_scale * bg_weight / (len(toks) * denom), 0.0, 0.0])
        if ref_boxes and ref_ca_loss_weight != 0.0:
            for o in range(self.n_obj):
                obj_boxes = bboxes[o]
                if not isinstance(obj_boxes[0], Iterable):
                    obj_boxes = [obj_boxes]
                if ref_ca_word_token_only:                   # guidance.py:213-219
                    toks = [word_token_indices[o]]
                elif ref_ca_last_token_only:
                    toks = [object_positions[o][-1]]
                else:
                    toks = object_positions[o]
                for bi, box in enumerate(obj_boxes):
                    for ki, key in enumerate(self.keys):
                        hw = map_hw[key]
                        Hs = int(math.sqrt(hw))
                        m = box_mask(box, Hs, Hs)
                        mid = len(masks)
                        masks.append(torch.nn.functional.pad(m.reshape(-1), (0, self.max_hw - hw)))
                        rid = len(self.ref_slots)
                        self.ref_slots.append((o, bi, ki))
                        for p in toks:
                            items.append([ki, 1, int(p), mid, 1, 1, rid, 0])
                            coefs.append([0.0, 0.0, loss_scale * ref_ca_loss_weight /
                                          (heads * len(obj_boxes) * len(toks) * denom), 0.0])
        self._host = (items, coefs, masks)
        self.n_samples = 1
        self._refs_host = None
        self._finalize()

    def _finalize(self):
        items, coefs, masks = self._host
        device, heads = self.device, self.heads
        # Items that write the same gradient column (map, image, token) are made adjacent and handled by one
        # workgroup per head in this fixed order: the sum of their contributions is reproducible bit for bit.
        order = sorted(range(len(items)), key=lambda i: (items[i][0], items[i][7], items[i][2], i))
        items, coefs = [items[i] for i in order], [coefs[i] for i in order]
        groups = []
        for i, it in enumerate(items):
            col = (it[0], it[7], it[2])
            if groups and groups[-1][2] == col:
                groups[-1][1] += 1
            else:
                groups.append([i, 1, col])
        self.n_groups = len(groups)
        self.groups = torch.tensor([g[:2] for g in groups] or [[0, 0]], dtype=torch.int32, device=device)
        self.n_items = len(items)
        self.items = torch.tensor(items if items else [[0] * 8], dtype=torch.int32, device=device)
        self.coefs = torch.tensor(coefs if coefs else [[0.0] * 4], dtype=F32, device=device)
        self.masks = (torch.stack(masks) if masks else torch.zeros(1, self.max_hw)).to(device, F32).contiguous()
        self.map_hw = torch.tensor([self._map_hw[k] for k in self.keys] or [1], dtype=torch.int32, device=device)
        self.partial = torch.zeros(max(self.n_items * heads, 1), dtype=F32, device=device)
        self.loss = torch.zeros(self.n_samples, dtype=F32, device=device)
        self.n_refs = len(self.ref_slots)
        self.refs = None        # fp32 [T][n_refs][heads][max_hw], filled by set_refs
        self._ptrs = None

    @classmethod
    def merged(cls, tables: "List[Optional[EnergyTables]]") -> "EnergyTables":
        """One table for a batch of images: image b's items carry sample index b (items[7]) and the
        kernel returns one loss per image.  `None` entries are images without guidance."""
        first = next(t for t in tables if t is not None)
        m = cls.__new__(cls)
        m.device, m.keys, m.heads, m.T, m.max_hw, m._map_hw = (first.device, first.keys, first.heads, first.T,
                                                                first.max_hw, first._map_hw)
        items, coefs, masks, m.ref_slots, refs = [], [], [], [], []
        for b, t in enumerate(tables):
            if t is None:
                continue
            assert t.keys == m.keys and t.max_hw == m.max_hw
            it, co, ma = t._host
            for row in it:
                r = list(row)
                r[3] += len(masks)
                r[6] += len(m.ref_slots) if r[1] == 1 else 0
                r[7] = b
                items.append(r)
            coefs += co
            masks += ma
            m.ref_slots += t.ref_slots
            if t.refs is not None:
                refs.append(t.refs)
        m.n_obj = sum(t.n_obj for t in tables if t is not None)
        m._host = (items, coefs, masks)
        m.n_samples = len(tables)
        m._finalize()
        if refs:
            m.set_refs(torch.cat(refs, dim=1))
        return m

    def bind(self, maps: Dict[Tuple, torch.Tensor], gmaps: Optional[Dict[Tuple, torch.Tensor]]):
        """Device pointer tables to the (static) map / map-gradient buffers of a guidance plan."""
        mp = torch.tensor([maps[k].data_ptr() for k in self.keys] or [0], dtype=torch.int64, device=self.device)
        gp = None
        if gmaps is not None:
            gp = torch.tensor([gmaps[k].data_ptr() for k in self.keys] or [0], dtype=torch.int64, device=self.device)
        self._ptrs = (mp, gp, maps, gmaps)

    def set_refs(self, refs: torch.Tensor):
        """refs: fp32 [T][n_refs][heads][max_hw] — stage-A maps R_b of guidance.py:201 per step."""
        assert refs.shape[1:] == (self.n_refs, self.heads, self.max_hw), refs.shape
        self.refs = refs.to(self.device, F32).contiguous()

    def run(self, dyn: torch.Tensor, grad_scale: float = 1.0, with_grad: bool = True) -> torch.Tensor:
        """Launches the energy (+ map gradients into the bound gmaps, zeroed here) for the step held in
        the device int32 `dyn[0]`.  Returns the device scalar loss (already multiplied by loss_scale,
        pipelines.py:48)."""
        mp, gp, maps, gmaps = self._ptrs
        if with_grad and gmaps is not None:
            for k in self.keys:
                ops.zero_(gmaps[k])
        stride = self.refs[0].numel() if self.refs is not None else 0
        ops.ca_energy(mp, gp if with_grad else None, self.map_hw, self.items, self.coefs, self.masks,
                      self.refs, stride, dyn, self.groups, self.n_groups, self.n_items, self.heads, self.T,
                      self.max_hw, self.partial, self.loss, grad_scale=grad_scale, n_samples=self.n_samples)
        return self.loss


# =================================================================================================
# BoxDiff (utils/boxdiff.py) — the energy of the `boxdiff` stage-2 baseline (generation/boxdiff.py)
# =================================================================================================
def gaussian_kernel(kernel_size=3, sigma=0.5) -> torch.Tensor:
    """utils/attn.py:92-110 (GaussianSmoothing, dim = 2): per axis 1 / (std sqrt(2 pi)) exp(-((x - mean) / (2 std))^2) —
    the (2 std) sits INSIDE the square there — multiplied over both axes, normalised to sum 1; fp32 as the reference."""
    ax = torch.arange(kernel_size, dtype=F32)
    kernel = torch.ones((kernel_size, kernel_size), dtype=F32)
    for mg in torch.meshgrid([ax, ax], indexing="ij"):
        mean = (kernel_size - 1) / 2
        kernel = kernel * (1 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-((mg - mean) / (2 * sigma)) ** 2))
    return kernel / torch.sum(kernel)


class BoxDiffTables:
    """Host side of the BoxDiff energy for one layout: masks, corner masks, projections, top-k counts and the item list
    that `lgd_boxdiff_energy_f32` consumes (utils/boxdiff.py:20-101).  Same interface as EnergyTables (`keys`, `bind`,
    `run`, `merged`), so the sampler's guidance loop drives either."""

    def __init__(self, device, bboxes, object_positions, guidance_attn_keys: Sequence[Tuple], map_hw: Dict[Tuple, int],
                 heads: int, text_len: int = 77, *, loss_scale=10.0, P=0.2, L=1, smooth_attentions=True, sigma=0.5,
                 kernel_size=3):
        self.device, self.heads, self.T = device, heads, text_len
        self.keys = [tuple(k) for k in guidance_attn_keys]
        hws = {map_hw[k] for k in self.keys}
        if len(hws) != 1:
            # compute_ca_loss_boxdiff concatenates the maps of all keys over heads and averages them (:152)
            raise RuntimeError(f"BoxDiff averages the maps of its guidance keys: they must share one resolution, got {sorted(hws)}")
        self.hw = hws.pop()
        self.side = int(math.sqrt(self.hw))
        if kernel_size != 3 and smooth_attentions:
            raise RuntimeError("BoxDiff smoothing: only the reference's 3x3 kernel is implemented")
        self.max_hw = self.hw
        self.loss_scale = float(loss_scale)
        self.n_obj = len(bboxes)
        self.smooth_host = gaussian_kernel(kernel_size, sigma).reshape(-1) if smooth_attentions else None
        S = self.side
        items, masks = [], []
        for o in range(self.n_obj):
            obj_boxes = bboxes[o]
            if not isinstance(obj_boxes[0], Iterable):
                obj_boxes = [obj_boxes]
            m = torch.zeros(S, S)
            cx, cy = torch.zeros(S), torch.zeros(S)
            for b in obj_boxes:
                x0, y0, x1, y1 = scale_proportion(b, S, S)
                m[y0:y1, x0:x1] = 1
                cx[max(x0 - L, 0):min(x0 + L + 1, S)] = 1.                    # utils/boxdiff.py:64-67
                cx[max(x1 - L, 0):min(x1 + L + 1, S)] = 1.
                cy[max(y0 - L, 0):min(y0 + L + 1, S)] = 1.
                cy[max(y1 - L, 0):min(y1 + L + 1, S)] = 1.
            k_fg = int((m.sum() * P).long())                                   # :80
            k_bg = int(((1 - m).sum() * P).long())                             # :85
            rows = torch.zeros(3, self.hw)
            rows[0] = m.reshape(-1)
            rows[1, :S], rows[1, S:2 * S] = cx, cy
            rows[2, :S], rows[2, S:2 * S] = m.max(dim=0).values, m.max(dim=1).values      # :90-91
            mid = len(masks)
            masks.append(rows)
            for p in object_positions[o]:
                if not 1 <= int(p) <= text_len - 2:
                    # the reference indexes attention_for_text[:, :, p - 1] of the [1:-1] slice (:46): p = 0 would wrap
                    raise RuntimeError(f"BoxDiff phrase token {p} outside 1..{text_len - 2}")
                items.append([int(p), mid, k_fg, k_bg, 0, 0, 0, 0])
        self._host = (items, masks)
        self.n_samples = 1
        self._groups_host = [[0, len(items)]]
        self._finalize()

    def _finalize(self):
        items, masks = self._host
        dev = self.device
        self.n_items = len(items)
        self.items = torch.tensor(items if items else [[0] * 8], dtype=torch.int32, device=dev)
        self.masks = (torch.stack(masks) if masks else torch.zeros(1, 3, self.hw)).to(dev, F32).contiguous()
        self.groups = torch.tensor(self._groups_host, dtype=torch.int32, device=dev)
        self.max_items = max(c for _, c in self._groups_host)
        self.smooth = self.smooth_host.to(dev, F32).contiguous() if self.smooth_host is not None else None
        self.loss = torch.zeros(self.n_samples, dtype=F32, device=dev)
        self.refs = None
        self._ptrs = None

    @classmethod
    def merged(cls, tables: "List[Optional[BoxDiffTables]]") -> "BoxDiffTables":
        """One table for a batch of images (None = an image without guidance: zero items, loss 0)."""
        first = next(t for t in tables if t is not None)
        m = cls.__new__(cls)
        for k in ("device", "heads", "T", "keys", "hw", "side", "max_hw", "loss_scale", "smooth_host"):
            setattr(m, k, getattr(first, k))
        items, masks, groups = [], [], []
        for t in tables:
            if t is None:
                groups.append([len(items), 0])
                continue
            assert t.keys == m.keys and t.hw == m.hw and t.loss_scale == m.loss_scale
            assert (t.smooth_host is None) == (m.smooth_host is None)
            it, ma = t._host
            groups.append([len(items), len(it)])
            for row in it:
                r = list(row)
                r[1] += len(masks)
                items.append(r)
            masks += ma
        m.n_obj = sum(t.n_obj for t in tables if t is not None)
        m._host, m._groups_host, m.n_samples = (items, masks), groups, len(tables)
        m._finalize()
        return m

    def bind(self, maps: Dict[Tuple, torch.Tensor], gmaps: Optional[Dict[Tuple, torch.Tensor]]):
        mp = torch.tensor([maps[k].data_ptr() for k in self.keys], dtype=torch.int64, device=self.device)
        gp = None
        if gmaps is not None:
            gp = torch.tensor([gmaps[k].data_ptr() for k in self.keys], dtype=torch.int64, device=self.device)
        self._ptrs = (mp, gp, maps, gmaps)

    def run(self, dyn: torch.Tensor = None, grad_scale: float = 1.0, with_grad: bool = True) -> torch.Tensor:
        """Launches the energy (+ map gradients into the bound gmaps, zeroed here: tokens 0 and T-1 get no gradient).
        Returns the device losses [n_samples], already multiplied by amp_loss_scale (utils/boxdiff.py:224)."""
        mp, gp, maps, gmaps = self._ptrs
        if with_grad and gmaps is not None:
            for k in self.keys:
                ops.zero_(gmaps[k])
        ops.boxdiff_energy(mp, gp if with_grad else None, len(self.keys), self.side, self.items, self.masks, self.smooth,
                           self.groups, self.n_samples, self.max_items, self.heads, self.T, self.loss_scale, grad_scale,
                           self.loss)
        return self.loss
