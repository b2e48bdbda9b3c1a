"""numpy restatement of the reference's label assignment and letterbox bookkeeping.

Test infrastructure only (see ``oracle/__init__.py``); parity unpinned (no fixtures in the
reference; its own code path needs cv2 and TensorFlow).

Follows
  WIDERDataset.preprocess_true_boxes   /root/reference/dataset.py:387-482
  image_preprocess (box arithmetic)    /root/reference/utils.py:16-31

The reference computes in float64 (np.zeros defaults) and calls utils.bbox_iou, a
tf.function that keeps float64 inputs in float64; so does this file.
"""
import numpy as np

MAX_BBOX_PER_SCALE = 128      # dataset.py:9


def _iou_xywh(b1, b2):
    """utils.py:33-63 on float64 arrays, broadcasting."""
    a1 = b1[..., 2] * b1[..., 3]
    a2 = b2[..., 2] * b2[..., 3]
    c1 = np.concatenate([b1[..., :2] - b1[..., 2:] * 0.5, b1[..., :2] + b1[..., 2:] * 0.5], -1)
    c2 = np.concatenate([b2[..., :2] - b2[..., 2:] * 0.5, b2[..., :2] + b2[..., 2:] * 0.5], -1)
    lu = np.maximum(c1[..., :2], c2[..., :2])
    rd = np.minimum(c1[..., 2:], c2[..., 2:])
    inter = np.maximum(rd - lu, 0.0)
    ia = inter[..., 0] * inter[..., 1]
    union = a1 + a2 - ia
    out = np.zeros_like(union)
    np.divide(ia, union, out=out, where=union != 0)
    return out


def scale_anchors(anchors, strides, literal=False):
    """Per-scale anchor sizes in grid units, [3][3][2].

    literal=False: the 18 configured pixel sizes reshaped to [3,3,2] and divided by the
    scale's stride, what the assignment evidently intends (boxes are compared in grid
    units).  literal=True: what dataset.py:51,440 does as written -- ``self.anchors`` stays
    the flat list of 18 numbers, so ``self.anchors[i]`` is ONE number and all three anchors
    of scale i get width = height = anchors_flat[i]."""
    flat = np.asarray(anchors, np.float64).reshape(-1)
    if literal:
        return np.stack([np.full((3, 2), flat[i]) for i in range(3)])
    return flat.reshape(3, 3, 2) / np.asarray(strides, np.float64)[:, None, None]


def preprocess_true_boxes(bboxes, input_size, num_classes, strides, anchors_grid):
    """bboxes [n,5] (xmin, ymin, xmax, ymax, class) -> (label_s, label_m, label_l, sboxes,
    mboxes, lboxes); dataset.py:387-482.  anchors_grid [3,3,2] from scale_anchors."""
    strides = np.asarray(strides, np.float64)
    sizes = [int(input_size // s) for s in strides]
    label = [np.zeros((sizes[i], sizes[i], 3, 5 + num_classes)) for i in range(3)]
    bboxes_xywh = [np.zeros((MAX_BBOX_PER_SCALE, 4)) for _ in range(3)]
    bbox_count = np.zeros((3,))
    for bbox in np.asarray(bboxes, np.float64):
        coor = bbox[:4]
        cls = int(bbox[4])
        # Build-defined: a box whose class is outside [0, C) or whose centre cell lies outside the
        # grid on any scale is skipped whole (the reference's numpy indexing raises IndexError, or
        # silently wraps a negative index); ymx_assign_labels does the same.
        ctr = (coor[2:] + coor[:2]) * 0.5
        cells = [np.floor(ctr / s) for s in strides]
        if not (0 <= cls < num_classes) or any((c < 0).any() or (c >= sizes[i]).any() for i, c in enumerate(cells)):
            continue
        onehot = np.zeros(num_classes)
        onehot[cls] = 1.0
        smooth = onehot * (1 - 0.01) + 0.01 * np.full(num_classes, 1.0 / num_classes)        # :419-423
        xywh = np.concatenate([(coor[2:] + coor[:2]) * 0.5, coor[2:] - coor[:2]], -1)
        scaled = 1.0 * xywh[None, :] / strides[:, None]                                       # :432
        iou = []
        exist_positive = False
        for i in range(3):
            anc = np.zeros((3, 4))
            anc[:, 0:2] = np.floor(scaled[i, 0:2]).astype(np.int32) + 0.5
            anc[:, 2:4] = anchors_grid[i]
            iou_scale = _iou_xywh(scaled[i][None, :], anc)
            iou.append(iou_scale)
            mask = iou_scale > 0.3
            if np.any(mask):
                xind, yind = np.floor(scaled[i, 0:2]).astype(np.int32)
                label[i][yind, xind, mask, :] = 0
                label[i][yind, xind, mask, 0:4] = xywh
                label[i][yind, xind, mask, 4:5] = 1.0
                label[i][yind, xind, mask, 5:] = smooth
                ind = int(bbox_count[i] % MAX_BBOX_PER_SCALE)
                bboxes_xywh[i][ind, :4] = xywh
                bbox_count[i] += 1
                exist_positive = True
        if not exist_positive:
            best = int(np.argmax(np.array(iou).reshape(-1), axis=-1))
            det, a = best // 3, best % 3
            xind, yind = np.floor(scaled[det, 0:2]).astype(np.int32)
            label[det][yind, xind, a, :] = 0
            label[det][yind, xind, a, 0:4] = xywh
            label[det][yind, xind, a, 4:5] = 1.0
            label[det][yind, xind, a, 5:] = smooth
            ind = int(bbox_count[det] % MAX_BBOX_PER_SCALE)
            bboxes_xywh[det][ind, :4] = xywh
            bbox_count[det] += 1
    return label[0], label[1], label[2], bboxes_xywh[0], bboxes_xywh[1], bboxes_xywh[2]


def letterbox_boxes(bboxes, height, width, train_input_size):
    """The box bookkeeping of image_preprocess (utils.py:16-31): scale by min(S/w, S/h), shift
    x by the width padding and y by the height padding.  Returns (boxes, scale, pad_x, pad_y)."""
    S = float(train_input_size)
    scale = min(S / width, S / height)
    new_h, new_w = int(scale * height), int(scale * width)
    pad_h, pad_w = (train_input_size - new_h) // 2, (train_input_size - new_w) // 2
    out = np.array(bboxes, np.float64, copy=True)
    out[:, [0, 2]] = out[:, [0, 2]] * scale + pad_w
    out[:, [1, 3]] = out[:, [1, 3]] * scale + pad_h
    return out, scale, pad_w, pad_h
