"""models/sam.py of the reference (mask refinement with SAM, :13-213): the same function names over
lgd_amd.sam_refine, whose network is the HIP implementation lgd_amd.sam.HipSamModel."""
from lgd_amd.sam_refine import (DeviceSamProcessor, SamRefiner, get_iou_with_resize, load_sam, preprocess_mask, sam, sam_box_input,  # noqa: F401
                                sam_point_input, sam_refine_attn, sam_refine_box, sam_refine_boxes, select_mask, wrap_sam)
