"""Mask refinement with SAM: the host rules of the reference's models/sam.py around the network
of lgd_amd.sam.HipSamModel.

Same functions and argument meaning as the reference module (models/sam.py:13-213, re-exported under that name by
dropin/models/sam.py); the model object in `sam_model_dict["sam_model"]` is the HIP implementation, the processor object
stays the Hugging Face `SamProcessor` (host-side image resizing / normalisation and mask post-processing).

  sam()               :25-55   processor -> model -> post_process_masks -> bilinear resize to the latent grid
  select_mask()       :67-111  "largest_over_conf": the largest of the three masks, candidates with low predicted IoU
                               or low IoU with the coarse mask pushed behind every admissible one
  sam_refine_boxes()  :182-213 LMD+: the layout box is the prompt and the coarse mask
  sam_refine_attn()   :125-172 LMD: the smoothed, thresholded cross-attention map gives a box (or its arg-max a point)

Reference quirks kept on purpose: `sam()` returns the predicted IoUs of image 0 / prompt 0 only (:45), and
`sam_refine_boxes` scores every box of every image with them (the plugins call it with one image and one box);
`sam_refine_attn(use_box_input=True)` hands the processor a two-level box list (:141-144), which transformers refuses
("Input boxes must be a list of list of list of floating points") — the plugins' default is the point prompt.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage

from . import hostprep
from .sam import HipSamModel, SamConfig


class _Encoding(dict):
    """What the processor returns, as far as models/sam.py:39 uses it: a mapping with `.to(device)`."""

    def to(self, *_a, **_k):
        return self


class DeviceSamProcessor:
    """The Hugging Face `SamProcessor` + `SamImageProcessor` defaults restated with torch ops on the model's device, for
    generation loops where the host-side PIL path (resize of every decoded single-object image on the CPU) would stall
    the GPU: longest edge -> 1024 (bilinear, half-pixel centres, result rounded to the 8-bit grid as PIL does), 1/255,
    ImageNet mean / std, zero padding bottom / right; prompts scaled by new/old size ([ext] processing_sam.py
    `_normalize_coordinates`); `post_process_masks` = interpolate to the padded size, crop, interpolate to the original
    size, threshold ([ext] image_processing_sam.py).  Differs from the PIL path by the 8-bit rounding of ties only."""
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def __init__(self, device="cuda", longest_edge=1024):
        self.dev, self.target = torch.device(device), int(longest_edge)
        self.image_processor = self
        self.on_device = True

    def _shape(self, h, w):
        scale = self.target / max(h, w)
        return int(h * scale + 0.5), int(w * scale + 0.5)

    def __call__(self, images, input_points=None, input_labels=None, input_boxes=None, return_tensors="pt", **_kw):
        imgs = [images] if isinstance(images, np.ndarray) and images.ndim == 3 else list(images)
        mean = torch.tensor(self.MEAN, device=self.dev).view(1, 3, 1, 1)
        std = torch.tensor(self.STD, device=self.dev).view(1, 3, 1, 1)
        px, orig, resh = [], [], []
        for im in imgs:
            t = torch.as_tensor(np.ascontiguousarray(im)).to(self.dev)
            if t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("images must be HxWx3 arrays")
            h, w = int(t.shape[0]), int(t.shape[1])
            nh, nw = self._shape(h, w)
            r = F.interpolate(t.permute(2, 0, 1)[None].float(), (nh, nw), mode="bilinear", align_corners=False,
                              antialias=(nh < h or nw < w)).round().clamp(0, 255)
            r = (r / 255.0 - mean) / std
            px.append(F.pad(r, (0, self.target - nw, 0, self.target - nh)))
            orig.append([h, w])
            resh.append([nh, nw])
        enc = _Encoding(pixel_values=torch.cat(px), original_sizes=torch.tensor(orig), reshaped_input_sizes=torch.tensor(resh))

        def scaled(prompts, box):
            if len(prompts) != len(imgs):
                sizes = [(orig[0], resh[0])] * len(prompts)
            else:
                sizes = list(zip(orig, resh))
            out = []
            for p, ((h, w), (nh, nw)) in zip(prompts, sizes):
                a = np.array(p, dtype=np.float64)
                a = a.reshape(-1, 2, 2) if box else a
                a[..., 0] *= nw / w
                a[..., 1] *= nh / h
                out.append(a.reshape(-1, 4) if box else a)
            return torch.from_numpy(np.array(out))
        if input_boxes is not None:
            if not (isinstance(input_boxes, list) and isinstance(input_boxes[0], list) and isinstance(input_boxes[0][0], list)):
                raise ValueError("Input boxes must be a list of list of list of floating points.")
            b = scaled(input_boxes, True)
            enc["input_boxes"] = b.unsqueeze(1) if b.dim() != 3 else b
        if input_points is not None:
            if not (isinstance(input_points, list) and isinstance(input_points[0], list)):
                raise ValueError("Input points must be a list of list of list of floating points.")
            pts = scaled(input_points, False)
            enc["input_points"] = pts.unsqueeze(1) if pts.dim() != 4 else pts
        if input_labels is not None:
            lab = torch.as_tensor(np.array(input_labels))
            enc["input_labels"] = lab.unsqueeze(1) if lab.dim() != 3 else lab
        return enc

    def post_process_masks(self, masks, original_sizes, reshaped_input_sizes, mask_threshold=0.0, binarize=True):
        out = []
        for i, (h, w) in enumerate(torch.as_tensor(original_sizes).tolist()):
            nh, nw = torch.as_tensor(reshaped_input_sizes).tolist()[i]
            m = F.interpolate(masks[i].float(), (self.target, self.target), mode="bilinear", align_corners=False)
            m = F.interpolate(m[..., :nh, :nw], (h, w), mode="bilinear", align_corners=False)
            out.append(m > mask_threshold if binarize else m)
        return out


def wrap_sam(hf_sam_model, sam_processor=None, device="cuda"):
    """A loaded Hugging Face `SamModel` (weights) -> the `sam_model_dict` of the reference with the HIP model in it."""
    if sam_processor is None:
        import transformers
        sam_processor = transformers.SamProcessor(transformers.SamImageProcessor())
    elif sam_processor == "device":
        sam_processor = DeviceSamProcessor(device)
    model = HipSamModel(SamConfig.from_hf(hf_sam_model.config), hf_sam_model.state_dict(), device=device)
    return dict(sam_model=model, sam_processor=sam_processor)


def load_sam(checkpoint="facebook/sam-vit-base", device="cuda"):
    """models/sam.py:13-21."""
    from transformers import SamModel, SamProcessor
    return wrap_sam(SamModel.from_pretrained(checkpoint), SamProcessor.from_pretrained(checkpoint), device)


def _listify(boxes):
    """Tuples anywhere in the nested box list -> lists (the processor only takes lists; :29-35)."""
    if isinstance(boxes, (tuple, list)):
        return [_listify(b) for b in boxes]
    return boxes


def sam(sam_model_dict, image, input_points=None, input_boxes=None, target_mask_shape=None, return_numpy=True):
    """-> (per image: bool masks [n_prompts, 3, h, w] at `target_mask_shape`, predicted IoUs of image 0 / prompt 0 [3])."""
    model, processor = sam_model_dict["sam_model"], sam_model_dict["sam_processor"]
    if input_boxes:
        input_boxes = _listify(input_boxes)
    enc = processor(image, input_points=input_points, input_boxes=input_boxes, return_tensors="pt")
    out = model(**enc)
    logits = out.pred_masks.float()
    if not getattr(processor, "on_device", False):                 # the Hugging Face processor works on host tensors
        logits = logits.cpu()
    full = processor.image_processor.post_process_masks(logits, enc["original_sizes"].cpu(), enc["reshaped_input_sizes"].cpu())
    conf_scores = out.iou_scores.float().cpu().numpy()[0, 0]
    small = [F.interpolate(m.float(), target_mask_shape, mode="bilinear").bool().cpu() for m in full]
    return ([m.numpy() for m in small] if return_numpy else small), conf_scores


def sam_point_input(sam_model_dict, image, input_points, **kwargs):
    return sam(sam_model_dict, image, input_points=input_points, **kwargs)


def sam_box_input(sam_model_dict, image, input_boxes, **kwargs):
    return sam(sam_model_dict, image, input_boxes=input_boxes, **kwargs)


def get_iou_with_resize(mask, masks, masks_shape):
    """IoU of `mask` with each candidate after bringing the candidates to `masks_shape` (:63-65: cv2.resize of the 0/255
    image, INTER_LINEAR, non-zero = inside; the plugins pass candidates that already have that shape)."""
    h, w = masks_shape
    cand = np.asarray(masks).astype(bool)
    if cand.shape[1:] != (h, w):
        t = torch.from_numpy(cand.astype(np.float32))[:, None]
        cand = (F.interpolate(t, (h, w), mode="bilinear", align_corners=False)[:, 0] > 0).numpy()
    return hostprep.iou(np.asarray(mask), cand)


def select_mask(masks, conf_scores, coarse_ious=None, rule="largest_over_conf", discourage_mask_below_confidence=0.85,
                discourage_mask_below_coarse_iou=0.2, verbose=False):
    if rule != "largest_over_conf":
        raise ValueError(f"Unknown rule: {rule}")
    area = masks.sum(axis=(1, 2))
    penalty = area.max()
    score = area - penalty * (conf_scores < discourage_mask_below_confidence)
    if coarse_ious is not None:
        score = score - penalty * (coarse_ious < discourage_mask_below_coarse_iou)
    best = int(np.argmax(score))
    if verbose:
        print(f"mask_sizes: {area}, scores: {score}")
        print(f"Selected a mask with confidence: {conf_scores[best]}, "
              f"coarse_iou: {None if coarse_ious is None else coarse_ious[best]}")
    return masks[best], conf_scores[best]


def preprocess_mask(token_attn_np_smooth, mask_th, n_erode_dilate_mask=0):
    """Min-max normalise, threshold, optionally open (erode then dilate) the binary map (:113-122)."""
    a = token_attn_np_smooth - token_attn_np_smooth.min()
    binary = a / a.max() > mask_th
    if n_erode_dilate_mask:
        binary = ndimage.binary_dilation(ndimage.binary_erosion(binary, iterations=n_erode_dilate_mask),
                                         iterations=n_erode_dilate_mask)
    return binary


def _pick(three_masks, conf_scores, coarse, below_conf, below_iou):
    ious = get_iou_with_resize(coarse, three_masks, masks_shape=coarse.shape)
    return select_mask(three_masks, conf_scores, coarse_ious=ious, rule="largest_over_conf",
                       discourage_mask_below_confidence=below_conf, discourage_mask_below_coarse_iou=below_iou, verbose=True)


def sam_refine_attn(sam_input_image, token_attn_np, model_dict, height, width, H, W, use_box_input, gaussian_sigma,
                    mask_th_for_box, n_erode_dilate_mask_for_box, mask_th_for_point, discourage_mask_below_confidence,
                    discourage_mask_below_coarse_iou, verbose):
    smooth = ndimage.gaussian_filter(token_attn_np.astype(float), sigma=gaussian_sigma)
    up_w, up_h = height // smooth.shape[1], width // smooth.shape[0]       # (w, h) order of the reference (:135)
    if use_box_input:
        coarse = preprocess_mask(smooth, mask_th_for_box, n_erode_dilate_mask=n_erode_dilate_mask_for_box)
        box = hostprep.binary_mask_to_box(coarse, w_scale=up_w, h_scale=up_h)
        masks, conf = sam_box_input(model_dict, image=sam_input_image, input_boxes=[box], target_mask_shape=(H, W))
    else:
        coarse = preprocess_mask(smooth, mask_th_for_point, n_erode_dilate_mask=0)
        peak_y, peak_x = np.unravel_index(smooth.argmax(), smooth.shape)
        masks, conf = sam_point_input(model_dict, image=sam_input_image, input_points=[[[peak_x * up_h, peak_y * up_w]]],
                                      target_mask_shape=(H, W))
    return _pick(masks[0][0], conf, coarse, discourage_mask_below_confidence, discourage_mask_below_coarse_iou)


def sam_refine_boxes(sam_input_images, boxes, model_dict, height, width, H, W, discourage_mask_below_confidence,
                     discourage_mask_below_coarse_iou, verbose):
    pixel_boxes = [[hostprep.scale_proportion(b, H=height, W=width) for b in per_image] for per_image in boxes]
    masks, conf = sam_box_input(model_dict, image=sam_input_images, input_boxes=pixel_boxes, target_mask_shape=(H, W))
    picked = [[_pick(three, conf, hostprep.proportion_to_mask(b, H, W, return_np=True), discourage_mask_below_confidence,
                     discourage_mask_below_coarse_iou) for b, three in zip(per_image, per_image_masks)]
              for per_image, per_image_masks in zip(boxes, masks)]
    return [[m for m, _ in row] for row in picked], [[c for _, c in row] for row in picked]


def sam_refine_box(sam_input_image, box, *args, **kwargs):
    masks, confs = sam_refine_boxes([sam_input_image], [[box]], *args, **kwargs)
    return masks[0][0], confs[0][0]


class SamRefiner:
    """What the generation pipelines (lgd_amd.pipeline) call per generated single-object image: the arguments the
    reference's plugins put into `sam_refine_kwargs` (generation/lmd_plus.py:320-327, generation/lmd.py:354-366), bound
    once."""

    def __init__(self, sam_model_dict, height=512, width=512, discourage_mask_below_confidence=0.85,
                 discourage_mask_below_coarse_iou=0.25, use_box_input=False, gaussian_sigma=None, mask_th_for_box=0.05,
                 n_erode_dilate_mask_for_box=1, mask_th_for_point=0.25, verbose=False):
        self.md, self.verbose = sam_model_dict, verbose
        self.common = dict(height=height, width=width, H=height // 8, W=width // 8,
                           discourage_mask_below_confidence=discourage_mask_below_confidence,
                           discourage_mask_below_coarse_iou=discourage_mask_below_coarse_iou)
        if gaussian_sigma is None:                               # generation/lmd.py:39-40,336-338
            gaussian_sigma = 0.1 if use_box_input else 1.5
        self.attn_kw = dict(use_box_input=use_box_input, gaussian_sigma=gaussian_sigma, mask_th_for_box=mask_th_for_box,
                            n_erode_dilate_mask_for_box=n_erode_dilate_mask_for_box, mask_th_for_point=mask_th_for_point)

    def box(self, image, box):
        """LMD+ (generation/lmd_plus.py:122-128) -> (bool mask [H/8, W/8], confidence)."""
        return sam_refine_box(sam_input_image=image, box=box, model_dict=self.md, verbose=self.verbose, **self.common)

    def attn(self, image, token_attn_np):
        """LMD (generation/lmd.py:141-147) -> (bool mask [H/8, W/8], confidence)."""
        return sam_refine_attn(sam_input_image=image, token_attn_np=token_attn_np, model_dict=self.md,
                               verbose=self.verbose, **self.attn_kw, **self.common)
