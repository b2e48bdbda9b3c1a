"""numpy fp32 restatement of the reference's head post-processing.

Test infrastructure only (see ``oracle/__init__.py``); parity unpinned.

Follows /root/reference/utils.py:
  bbox_iou        utils.py:33-63
  bbox_giou       utils.py:65-105
  decode          utils.py:107-126
  decode_train    utils.py:128-131
  decode_predict  utils.py:133-140
  filter_bboxes   utils.py:142-167
  bbox_nms        utils.py:169-194  (call parameters only; the NMS arithmetic
                  is tf.image.combined_non_max_suppression, not in the repo)

Every array stays float32 so that results are reproducible operation by
operation on any IEEE-754 machine; the HIP NMS kernel repeats the IoU
arithmetic below in the same order, without fused multiply-add.
"""
import numpy as np

F32 = np.float32

# /root/reference/config.yml:3-5
ANCHORS = np.array([12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146,
                    142, 110, 192, 243, 459, 401], dtype=F32).reshape(3, 3, 2)
STRIDES = (8, 16, 32)
XYSCALE = (1.2, 1.1, 1.05)


def _sigmoid(x):
    x = np.asarray(x, dtype=F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def _divide_no_nan(a, b):
    out = np.zeros(np.broadcast(a, b).shape, dtype=F32)
    np.divide(a, b, out=out, where=(b != 0))
    return out


def _corners(b):
    return np.concatenate([b[..., :2] - b[..., 2:] * F32(0.5),
                           b[..., :2] + b[..., 2:] * F32(0.5)], axis=-1)


def bbox_iou(bboxes1, bboxes2):
    """IoU of centre-format (cx, cy, w, h) boxes, broadcasting; utils.py:33-63."""
    b1 = np.asarray(bboxes1, dtype=F32)
    b2 = np.asarray(bboxes2, dtype=F32)
    a1 = b1[..., 2] * b1[..., 3]
    a2 = b2[..., 2] * b2[..., 3]
    c1, c2 = _corners(b1), _corners(b2)
    left_up = np.maximum(c1[..., :2], c2[..., :2])
    right_down = np.minimum(c1[..., 2:], c2[..., 2:])
    inter = np.maximum(right_down - left_up, F32(0.0))
    inter_area = inter[..., 0] * inter[..., 1]
    union = a1 + a2 - inter_area
    return _divide_no_nan(inter_area, union)


def bbox_giou(bboxes1, bboxes2):
    """Generalised IoU of centre-format boxes; utils.py:65-105."""
    b1 = np.asarray(bboxes1, dtype=F32)
    b2 = np.asarray(bboxes2, dtype=F32)
    a1 = b1[..., 2] * b1[..., 3]
    a2 = b2[..., 2] * b2[..., 3]
    c1, c2 = _corners(b1), _corners(b2)
    left_up = np.maximum(c1[..., :2], c2[..., :2])
    right_down = np.minimum(c1[..., 2:], c2[..., 2:])
    inter = np.maximum(right_down - left_up, F32(0.0))
    inter_area = inter[..., 0] * inter[..., 1]
    union = a1 + a2 - inter_area
    iou = _divide_no_nan(inter_area, union)
    enc_lu = np.minimum(c1[..., :2], c2[..., :2])
    enc_rd = np.maximum(c1[..., 2:], c2[..., 2:])
    enc = enc_rd - enc_lu
    enc_area = enc[..., 0] * enc[..., 1]
    return iou - _divide_no_nan(enc_area - union, enc_area)


def decode(conv_output, scale, output_size, num_classes, strides=STRIDES,
           xyscale=XYSCALE, anchors=ANCHORS):
    """conv_output [B,S,S,3,5+C] -> (xywh [B,S,S,3,4] px, conf [...,1], prob [...,C]).

    utils.py:107-126.  tf.meshgrid default indexing is 'xy', so the grid value
    at [row, col] is (x=col, y=row) (utils.py:112-114).  Anchors are in
    pixels and are not multiplied by the stride (utils.py:120).
    """
    x = np.asarray(conv_output, dtype=F32)
    S = int(output_size)
    assert x.shape[1] == S and x.shape[2] == S and x.shape[3] == 3
    assert x.shape[4] == 5 + num_classes
    dxdy, dwdh, rconf, rprob = x[..., 0:2], x[..., 2:4], x[..., 4:5], x[..., 5:]
    gx, gy = np.meshgrid(np.arange(S), np.arange(S))      # 'xy' indexing
    grid = np.stack([gx, gy], axis=-1)[:, :, None, :].astype(F32)   # [S,S,1,2]
    xs = F32(xyscale[scale])
    stride = F32(strides[scale])
    anc = np.asarray(anchors, dtype=F32).reshape(3, 3, 2)[scale]    # [3,2]
    pred_xy = ((_sigmoid(dxdy) * xs) - F32(0.5) * (xs - F32(1.0)) + grid[None]) * stride
    pred_wh = np.exp(dwdh) * anc
    pred_xywh = np.concatenate([pred_xy, pred_wh], axis=-1).astype(F32)
    return pred_xywh, _sigmoid(rconf), _sigmoid(rprob)


def decode_train(conv_output, scale, output_size, num_classes, strides=STRIDES,
                 xyscale=XYSCALE, anchors=ANCHORS):
    """utils.py:128-131."""
    xywh, conf, prob = decode(conv_output, scale, output_size, num_classes,
                              strides, xyscale, anchors)
    return np.concatenate([xywh, conf, prob], axis=-1)


def decode_predict(conv_output, scale, output_size, num_classes, strides=STRIDES,
                   xyscale=XYSCALE, anchors=ANCHORS):
    """-> xywh [B,3S^2,4], score [B,3S^2,C]; flatten order (row, col, anchor).

    utils.py:133-140.
    """
    xywh, conf, prob = decode(conv_output, scale, output_size, num_classes,
                              strides, xyscale, anchors)
    B = xywh.shape[0]
    score = (conf * prob).astype(F32)
    return xywh.reshape(B, -1, 4), score.reshape(B, -1, num_classes)


def decode_all(heads, num_classes, strides=STRIDES, xyscale=XYSCALE, anchors=ANCHORS):
    """The intended composition of bbox_nms's loop (utils.py:172-175).

    The reference passes 3 of decode_predict's 7 arguments and concatenates on
    the last axis; this is the corrected form: decode every scale with its real
    arguments and concatenate on the candidate axis.  heads = (sbbox, mbbox,
    lbbox), each [B,S_i,S_i,3,5+C] (or the raw conv map [B,S_i,S_i,3(5+C)]).
    """
    boxes, scores = [], []
    for i, h in enumerate(heads):
        h = np.asarray(h, dtype=F32)
        if h.ndim == 4:
            h = h.reshape(h.shape[0], h.shape[1], h.shape[2], 3, 5 + num_classes)
        b, s = decode_predict(h, i, h.shape[1], num_classes, strides, xyscale, anchors)
        boxes.append(b)
        scores.append(s)
    return np.concatenate(boxes, axis=1), np.concatenate(scores, axis=1)


def filter_bboxes(bboxes, scores, score_threshold, image_size):
    """Per-image threshold compaction + xywh -> (ymin,xmin,ymax,xmax)/image_size.

    utils.py:142-167.  keep = max_c score >= threshold (>=, utils.py:146);
    survivors keep their original order (tf.boolean_mask); no clipping.  The
    reference flattens the batch before reshaping, which is only defined for
    B = 1; here every image is compacted on its own and padded with zeros to
    the largest survivor count.

    Returns boxes [B,Mmax,4], scores [B,Mmax,C], counts [B] int32,
    src_idx [B,Mmax] int32 (index into the unfiltered list, -1 padding).
    """
    bboxes = np.asarray(bboxes, dtype=F32)
    scores = np.asarray(scores, dtype=F32)
    B, N, C = scores.shape
    thr = F32(score_threshold)
    size = F32(image_size)
    keep = scores.max(axis=-1) >= thr if C > 0 else np.zeros((B, N), bool)
    counts = keep.sum(axis=1).astype(np.int32)
    mmax = int(counts.max()) if B > 0 else 0
    out_b = np.zeros((B, mmax, 4), F32)
    out_s = np.zeros((B, mmax, C), F32)
    out_i = np.full((B, mmax), -1, np.int32)
    for b in range(B):
        idx = np.nonzero(keep[b])[0]
        m = idx.size
        xy, wh = bboxes[b, idx, 0:2], bboxes[b, idx, 2:4]
        yx, hw = xy[:, ::-1], wh[:, ::-1]
        mins = (yx - (hw / F32(2.0))) / size
        maxs = (yx + (hw / F32(2.0))) / size
        out_b[b, :m, 0:2] = mins
        out_b[b, :m, 2:4] = maxs
        out_s[b, :m] = scores[b, idx]
        out_i[b, :m] = idx
    return out_b, out_s, counts, out_i


def nms_iou(box, others):
    """IoU of one corner box against many, the arithmetic the NMS uses.

    Corners may be flipped: each box is normalised with min/max first.  A box
    of area <= 0 has IoU 0 with everything.  fp32 throughout, operations in
    exactly this order (the HIP kernel mirrors it).
    """
    box = np.asarray(box, dtype=F32)
    others = np.asarray(others, dtype=F32).reshape(-1, 4)
    y1a, x1a = np.minimum(box[0], box[2]), np.minimum(box[1], box[3])
    y2a, x2a = np.maximum(box[0], box[2]), np.maximum(box[1], box[3])
    y1b, x1b = np.minimum(others[:, 0], others[:, 2]), np.minimum(others[:, 1], others[:, 3])
    y2b, x2b = np.maximum(others[:, 0], others[:, 2]), np.maximum(others[:, 1], others[:, 3])
    area_a = (y2a - y1a) * (x2a - x1a)
    area_b = (y2b - y1b) * (x2b - x1b)
    iy1, ix1 = np.maximum(y1a, y1b), np.maximum(x1a, x1b)
    iy2, ix2 = np.minimum(y2a, y2b), np.minimum(x2a, x2b)
    inter = np.maximum(iy2 - iy1, F32(0.0)) * np.maximum(ix2 - ix1, F32(0.0))
    union = (area_a + area_b) - inter
    iou = np.zeros(others.shape[0], F32)
    ok = (area_a > 0) & (area_b > 0)
    np.divide(inter, union, out=iou, where=ok)
    return iou


def combined_nms(boxes, scores, counts=None, max_output_size_per_class=32,
                 max_total_size=32, iou_threshold=0.45, score_threshold=0.25,
                 src_idx=None):
    """Combined (per-class, then per-image top-k) NMS; the semantics this build fixes.

    Stands for tf.image.combined_non_max_suppression as called at
    utils.py:184-192 (boxes [B,M,1,4] shared by all classes, pad_per_class
    False, clip_boxes True).  TensorFlow is absent, so the tie-breaks are
    defined here:

    per (image, class): candidates = score > score_threshold (strict); visited
      in descending score, equal scores by ascending box index; a candidate is
      kept iff its IoU with every box already kept for that class is
      <= iou_threshold; stop after max_output_size_per_class kept.
    per image: all classes' kept entries sorted by descending score, then
      ascending class id, then ascending box index; the first max_total_size
      survive; coordinates clipped to [0,1]; unused slots are zero.

    boxes [B,M,4] (ymin,xmin,ymax,xmax), scores [B,M,C], counts [B] (valid
    candidates per image; None = all M).  Returns boxes [B,T,4], scores [B,T],
    classes [B,T] float32, valid [B] int32, keep_idx [B,T] int32 (index into
    the candidate list, mapped through src_idx when given; -1 padding).
    """
    boxes = np.asarray(boxes, dtype=F32)
    scores = np.asarray(scores, dtype=F32)
    B, M, C = scores.shape
    T = int(max_total_size)
    thr_s = F32(score_threshold)
    thr_i = F32(iou_threshold)
    out_b = np.zeros((B, T, 4), F32)
    out_s = np.zeros((B, T), F32)
    out_c = np.zeros((B, T), F32)
    out_v = np.zeros((B,), np.int32)
    out_k = np.full((B, T), -1, np.int32)
    for b in range(B):
        m = M if counts is None else int(counts[b])
        merged = []                                    # (score, class, idx)
        for c in range(C):
            s = scores[b, :m, c]
            cand = np.nonzero(s > thr_s)[0]
            if cand.size == 0:
                continue
            order = cand[np.lexsort((cand, -s[cand]))]   # score desc, idx asc
            kept = []
            for i in order:
                if kept:
                    iou = nms_iou(boxes[b, i], boxes[b, kept])
                    if np.any(iou > thr_i):
                        continue
                kept.append(int(i))
                if len(kept) >= max_output_size_per_class:
                    break
            merged.extend((s[i], c, i) for i in kept)
        merged.sort(key=lambda t: (-float(t[0]), t[1], t[2]))
        merged = merged[:T]
        out_v[b] = len(merged)
        for j, (sc, c, i) in enumerate(merged):
            out_b[b, j] = np.clip(boxes[b, i], F32(0.0), F32(1.0))
            out_s[b, j] = sc
            out_c[b, j] = F32(c)
            out_k[b, j] = i if src_idx is None else int(src_idx[b, i])
    return out_b, out_s, out_c, out_v, out_k


def bbox_nms(heads, image_size, num_classes, max_outputs_per_class=32, max_outputs=32,
             iou_threshold=0.45, score_threshold=0.25, strides=STRIDES,
             xyscale=XYSCALE, anchors=ANCHORS):
    """decode every scale -> filter -> combined NMS (corrected utils.py:169-194).

    Returns (boxes [B,T,4], scores [B,T], classes [B,T], valid [B], keep_idx [B,T]);
    keep_idx indexes the unfiltered candidate list.
    """
    xywh, sc = decode_all(heads, num_classes, strides, xyscale, anchors)
    fb, fs, counts, src = filter_bboxes(xywh, sc, score_threshold, image_size)
    return combined_nms(fb, fs, counts, max_outputs_per_class, max_outputs,
                        iou_threshold, score_threshold, src_idx=src)
