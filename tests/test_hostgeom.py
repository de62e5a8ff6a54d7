"""Host-side box / mask geometry rules (pixel rounding of boxes, bounding boxes and mass centres of masks,
8x8-grid quantised shifts) against known-answer vectors produced by the
reference's own utils/utils.py on seeded random inputs (oracle/make_golden_hostgeom.py).  Integer / copy
work: exact; the two float outputs (centred boxes, mass centres) must be bit-identical too, because they
feed `round()` decisions downstream."""
import os

import numpy as np
import torch

import lgd_amd  # noqa: F401
from lgd_amd import hostprep as hp

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hostgeom.npz"))


def test_box_to_pixel_rect_and_mask():
    boxes = G["boxes"]
    for hw in (64, 32, 16, 8):
        got = np.array([hp.scale_proportion(list(b), hw, hw) for b in boxes], dtype=np.int64)
        assert np.array_equal(got, G[f"rect_{hw}"])
        got = np.array([hp.scale_proportion(list(b), hw, hw, use_legacy=True) for b in boxes], dtype=np.int64)
        assert np.array_equal(got, G[f"rect_legacy_{hw}"])
    m = np.stack([hp.proportion_to_mask(list(b), 64, 64).numpy() for b in boxes[:20]])
    assert np.array_equal(m, G["mask_64_first20"])
    assert np.array_equal(hp.proportion_to_mask(list(boxes[3]), 64, 64, return_np=True), G["mask_64_first20"][3])


def test_centered_boxes():
    boxes = G["boxes"]
    f = hp.get_centered_box
    assert np.array_equal(np.array([f(list(b), horizontal_center_only=True) for b in boxes]), G["centered_h"])
    assert np.array_equal(np.array([f(list(b), horizontal_center_only=False) for b in boxes]), G["centered_c"])
    assert np.array_equal(np.array([f(list(b), horizontal_center_only=False, vertical_center=0.3) for b in boxes]),
                          G["centered_c3"])
    assert np.array_equal(np.array([f(list(b), horizontal_center_only=False, vertical_placement="floor_padding",
                                      floor_padding=0.2) for b in boxes]), G["centered_f"])


def test_mask_bounding_box_box_mask_and_centre():
    for i in range(60):
        H, W = (int(v) for v in G["mask_hw"][i])
        m = torch.from_numpy(G["masks"][i][:H, :W].copy())
        assert [int(v) for v in hp.binary_mask_to_box(m)] == list(G["bbox_enlarged"][i])
        assert [int(v) for v in hp.binary_mask_to_box(m, enlarge_box_by_one=False, w_scale=2, h_scale=3)] == \
            list(G["bbox_plain_scaled"][i])
        assert np.array_equal(hp.binary_mask_to_box_mask(m).numpy(), G["box_masks"][i][:H, :W])
        assert tuple(hp.binary_mask_to_center(m)) == tuple(G["centers"][i])
        assert tuple(hp.binary_mask_to_center(m, normalize=True)) == tuple(G["centers_norm"][i])
        cn = hp.binary_mask_to_center(m.numpy(), normalize=True)               # numpy masks (SAM path)
        assert np.allclose(cn, G["centers_norm"][i], rtol=0, atol=1e-6)
    assert np.array_equal(hp.iou(G["masks"][0], G["masks"][1:10]), G["iou_0_vs_1to9"])


def test_quantised_shifts_with_zero_fill():
    lat, att = torch.from_numpy(G["shift_lat"]), torch.from_numpy(G["shift_att"])
    msk = torch.from_numpy(G["masks"][0].copy())
    for i, (dx, dy) in enumerate(G["shift_offsets"]):
        dx, dy = float(dx), float(dy)
        assert np.array_equal(hp.shift_tensor(lat, dx, dy, offset_normalized=True).numpy(), G[f"shift_lat_{i}"])
        assert np.array_equal(hp.shift_tensor(att, dx, dy, offset_normalized=True, ignore_last_dim=True).numpy(),
                              G[f"shift_att_{i}"])
        out = hp.shift_tensor(msk, dx, dy, offset_normalized=True)
        assert out.dtype == torch.bool and np.array_equal(out.numpy(), G[f"shift_msk_{i}"])
    for i, (dx, dy) in enumerate(G["shift_px_offsets"]):
        assert np.array_equal(hp.shift_tensor(lat, int(dx), int(dy)).numpy(), G[f"shift_px_{i}"])
    assert np.array_equal(np.array(hp.expand_overall_bboxes([[[0.1, 0.2, 0.3, 0.4]],
                                                             [[0.5, 0.5, 0.6, 0.7], [0.0, 0.1, 0.2, 0.3]]])), G["expand"])
