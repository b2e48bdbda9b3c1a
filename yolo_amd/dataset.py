"""Training-target construction, after the reference's dataset.py / utils.image_preprocess.

Of /root/reference/dataset.py the parts that feed the loss are provided:
``preprocess_true_boxes`` (dataset.py:387-482), batched and on the GPU, the target packing
of dataset.py:210-218 and the TFRecord shard writer of dataset.py:64-126 (``to_tfrecord``,
host-side, in yolo_amd/tfrecord.py).  The WIDER-face index parser and the OpenCV
augmentations are not provided.  ``utils.image_preprocess`` (utils.py:5-31) lives in
yolo_amd/utils.py as in the reference; ``letterbox_boxes`` here is its box arithmetic alone.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .config import config as _config
from .tfrecord import to_tfrecord            # noqa: F401  (dataset.py:64-126)

_YOLO = _config["yolo"]
MAX_BBOX_PER_SCALE = 128       # dataset.py:9


def scale_anchors(anchors=None, strides=None, literal=False):
    """Anchor sizes per scale in grid units, float32 [3,3,2].

    Default: the configured pixel sizes reshaped to [3,3,2] and divided by the scale's
    stride (boxes are compared with anchors in grid units, dataset.py:432-444).
    literal=True reproduces dataset.py:51,440 as written: ``self.anchors`` stays the flat
    list, ``self.anchors[i]`` is one number, and all three anchors of scale i are squares of
    that many grid cells."""
    flat = np.asarray(_YOLO["anchors"] if anchors is None else anchors, np.float64).reshape(-1)
    st = np.asarray(_YOLO["strides"] if strides is None else strides, np.float64)
    if literal:
        return np.stack([np.full((3, 2), flat[i]) for i in range(3)]).astype(np.float32)
    return (flat.reshape(3, 3, 2) / st[:, None, None]).astype(np.float32)


def preprocess_true_boxes(bboxes, counts, input_size, num_classes, strides=None, anchors_grid=None):
    """Batched label assignment on the GPU (ymx_assign_labels).

    bboxes [B,NMAX,5] float32 GPU tensor (xmin, ymin, xmax, ymax, class), counts [B] int32
    (boxes per image; None = NMAX each).  Returns (label_sbbox, label_mbbox, label_lbbox,
    sbboxes, mbboxes, lbboxes): labels [B,S_i,S_i,3,5+C], boxes [B,128,4], float32, as the
    reference builds them per image (dataset.py:396-482)."""
    _lib.require_gpu(bboxes, counts)
    if bboxes.dim() != 3 or bboxes.shape[2] != 5:
        raise ValueError("bboxes: expected [B,NMAX,5], got %s" % (tuple(bboxes.shape),))
    strides = list(_YOLO["strides"] if strides is None else strides)
    if anchors_grid is None:
        anchors_grid = scale_anchors(strides=strides)
    anc = np.ascontiguousarray(anchors_grid, np.float32).reshape(-1)
    if anc.size != 18 or len(strides) != 3:
        raise ValueError("three scales with three (w,h) anchors each are required")
    bx = bboxes.float().contiguous()
    B, nmax = bx.shape[0], bx.shape[1]
    cn = None if counts is None else counts.to(torch.int32).contiguous()
    dev = bx.device
    D = 5 + int(num_classes)
    sizes = [int(input_size) // int(s) for s in strides]
    labels = [torch.empty((B, s, s, 3, D), device=dev, dtype=torch.float32) for s in sizes]
    gts = [torch.empty((B, MAX_BBOX_PER_SCALE, 4), device=dev, dtype=torch.float32) for _ in sizes]
    lp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in labels])
    gp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in gts])
    st = (ctypes.c_float * 3)(*[float(s) for s in strides])
    ag = (ctypes.c_float * 18)(*[float(v) for v in anc])
    _lib.launch("ymx_assign_labels", bx, _lib.ptr(bx), _lib.ptr(cn), B, nmax, int(input_size), int(num_classes), st, ag, lp, gp)
    return labels[0], labels[1], labels[2], gts[0], gts[1], gts[2]


def pack_targets(label_s, label_m, label_l, sboxes, mboxes, lboxes):
    """The three per-scale targets [B, S*S*3*(5+C) + 128*4] the loss closures take
    (dataset.py:210-218)."""
    out = []
    for lab, bx in ((label_s, sboxes), (label_m, mboxes), (label_l, lboxes)):
        b = lab.shape[0]
        out.append(torch.cat([lab.reshape(b, -1), bx.reshape(b, -1)], dim=-1))
    return tuple(out)


def letterbox_boxes(bboxes, height, width, train_input_size):
    """Box arithmetic of image_preprocess (utils.py:19-31): boxes (xmin, ymin, xmax, ymax, ...)
    of a height x width image -> the letterboxed square of side train_input_size.  Host
    arithmetic on a small array; returns (boxes, scale, pad_x, pad_y)."""
    s = float(train_input_size)
    scale = min(s / width, s / height)
    new_h, new_w = int(scale * height), int(scale * width)
    pad_h, pad_w = (int(train_input_size) - new_h) // 2, (int(train_input_size) - new_w) // 2
    out = np.array(bboxes, np.float64, copy=True)
    out[:, [0, 2]] = out[:, [0, 2]] * scale + pad_w
    out[:, [1, 3]] = out[:, [1, 3]] * scale + pad_h
    return out, scale, pad_w, pad_h
