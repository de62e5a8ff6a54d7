"""ORACLE TEST INFRASTRUCTURE — stand-in for `cv2` (models/sam.py:10).  The reference only calls `cv2.resize` from
`get_iou_with_resize` (models/sam.py:63-65), and both plugins pass candidates that already have the target shape, where
INTER_LINEAR resizing is the identity; anything else is refused rather than approximated."""
import numpy as np

INTER_LINEAR = 1


def resize(src, dsize, interpolation=INTER_LINEAR):
    w, h = dsize
    if src.shape[:2] != (h, w):
        raise NotImplementedError("cv2 stand-in: only same-size resize (identity) is available in the sandbox")
    return np.array(src, copy=True)
