"""Head post-processing and box maths, same names and signatures as the reference's utils.py.

Mirrors /root/reference/utils.py:5-242.  image_preprocess and draw_bbox, OpenCV on numpy
arrays in the reference, take GPU tensors here (letterbox resize and box outlines are
kernels of libymx; text labels are not rendered).  All device work goes through libymx;
tensors must be on a GPU.

Layout note: score matrices produced here are stored class-major ([B,C,N] in memory)
because the NMS kernel streams one class row at a time; they are returned as [B,N,C]
*views* (``.transpose(1, 2)``), so indexing and shapes match the reference.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .config import config as _config

_YOLO = _config["yolo"]


def default_anchors():
    return np.asarray(_YOLO["anchors"], dtype=np.float32).reshape(3, 3, 2)


def _anchors_array(anchors):
    if isinstance(anchors, torch.Tensor):
        anchors = anchors.detach().cpu().numpy()
    a = np.ascontiguousarray(anchors, dtype=np.float32)
    if a.size % 6:
        raise ValueError("anchors must hold 3 (w,h) pairs per scale")
    return a.reshape(-1, 3, 2)


def _as_5d(conv_output, num_classes):
    """[B,S,S,3(5+C)] or [B,S,S,3,5+C] -> contiguous 5-D tensor (a view when possible)."""
    t = conv_output
    _lib.require_gpu(t)
    if t.dim() == 4:
        if t.shape[-1] != 3 * (5 + num_classes):
            raise ValueError("conv map has %d channels, expected 3*(5+%d)" % (t.shape[-1], num_classes))
        t = t.reshape(t.shape[0], t.shape[1], t.shape[2], 3, 5 + num_classes)
    if t.dim() != 5 or t.shape[3] != 3 or t.shape[4] != 5 + num_classes or t.shape[1] != t.shape[2]:
        raise ValueError("expected [B,S,S,3,5+C], got %s" % (tuple(t.shape),))
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    return t.contiguous()


def _round4(n):
    return (n + 3) // 4 * 4


class _Scales(object):
    """Host-side argument block of the decode entry points."""

    def __init__(self, heads, strides, xyscale, anchors, cell_ld=None):
        n = len(heads)
        self.n = n
        # cell_ld: elements from one grid cell to the next when the maps are padded views
        # (then ymx_decode_filter_strided is the entry point to call), else None
        self.cell_ld = None if cell_ld is None else (ctypes.c_int * n)(*[int(v) for v in cell_ld])
        self.ptrs = (ctypes.c_void_p * n)(*[h.data_ptr() for h in heads])
        self.S = (ctypes.c_int * n)(*[int(h.shape[1]) for h in heads])
        self.strides = (ctypes.c_float * n)(*[float(s) for s in strides])
        self.xyscale = (ctypes.c_float * n)(*[float(s) for s in xyscale])
        flat = np.ascontiguousarray(anchors, np.float32).reshape(-1)
        self.anchors = (ctypes.c_float * (6 * n))(*[float(v) for v in flat[:6 * n]])
        self.N = sum(3 * int(h.shape[1]) ** 2 for h in heads)


def decode_heads(heads, num_classes, strides, xyscale, anchors, filter=False, score_threshold=0.0,
                 image_size=1.0, split=False):
    """Decode consecutive scales in one launch (ymx_decode_filter / ymx_decode_split).

    heads: list of [B,S_i,S_i,3,5+C] tensors (same dtype); strides/xyscale: one value per
    head; anchors: [len(heads),3,2] pixels.  Returns a dict with boxes [B,cap,4],
    scores_cm [B,C,cap] (class-major), counts [B] int32, src_idx [B,cap] (filter only),
    conf [B,cap] (split only); cap = number of candidates rounded up to 4.
    """
    heads = [_as_5d(h, num_classes) for h in heads]
    dt = heads[0].dtype
    dev = heads[0].device
    if any(h.dtype != dt or h.shape[0] != heads[0].shape[0] for h in heads):
        raise ValueError("all scales must share batch size and dtype")
    B = heads[0].shape[0]
    sc = _Scales(heads, strides, xyscale, anchors)
    cap = _round4(sc.N)
    boxes = torch.empty((B, cap, 4), device=dev, dtype=torch.float32)
    scores = torch.empty((B, num_classes, cap), device=dev, dtype=torch.float32)
    counts = torch.empty((B,), device=dev, dtype=torch.int32)
    out = {"boxes": boxes, "scores_cm": scores, "counts": counts, "N": sc.N, "cap": cap, "_keep": heads}
    L = _lib.lib()
    if split:
        conf = torch.empty((B, cap), device=dev, dtype=torch.float32)
        _lib.launch("ymx_decode_split", boxes, sc.ptrs, sc.n, sc.S, B, num_classes, _lib.dtype_code(dt), sc.anchors,
                                      sc.strides, sc.xyscale, _lib.ptr(boxes), _lib.ptr(conf), _lib.ptr(scores),
                                      _lib.ptr(counts), cap)
        out["conf"] = conf
        return out
    src_idx = ws = None
    ws_bytes = 0
    if filter:
        src_idx = torch.empty((B, cap), device=dev, dtype=torch.int32)
        ws_bytes = L.ymx_decode_workspace_bytes(B, sc.n, sc.S)
        ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
        out["src_idx"] = src_idx
        out["workspace"] = ws
    _lib.launch("ymx_decode_filter", boxes, sc.ptrs, sc.n, sc.S, B, num_classes, _lib.dtype_code(dt), sc.anchors,
                                   sc.strides, sc.xyscale, 1 if filter else 0, float(score_threshold),
                                   float(image_size), _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(src_idx),
                                   _lib.ptr(counts), cap, _lib.ptr(ws), ws_bytes)
    return out


def decode(conv_output, scale, output_size, num_classes, strides, xyscale, anchors):
    """[B,S,S,3,5+C] -> (pred_xywh [B,S,S,3,4], pred_conf [B,S,S,3,1], pred_prob [B,S,S,3,C]).

    utils.py:107-126: xy = (sigmoid(t_xy)*xyscale[scale] - 0.5*(xyscale[scale]-1) + grid) *
    strides[scale] with grid[row, col] = (col, row); wh = exp(t_wh)*anchors[scale] (pixels);
    conf and prob are sigmoids.
    """
    t = _as_5d(conv_output, num_classes)
    B, S = t.shape[0], t.shape[1]
    if int(output_size) != S:
        raise ValueError("output_size %s does not match the conv map (%d)" % (output_size, S))
    anc = _anchors_array(anchors)[scale]
    r = decode_heads([t], num_classes, [strides[scale]], [xyscale[scale]], anc[None], split=True)
    n = r["N"]
    xywh = r["boxes"][:, :n].reshape(B, S, S, 3, 4)
    conf = r["conf"][:, :n].reshape(B, S, S, 3, 1)
    prob = r["scores_cm"][:, :, :n].transpose(1, 2).reshape(B, S, S, 3, num_classes)
    return xywh, conf, prob


def decode_train(conv_output, scale, output_size, num_classes, strides, xyscale, anchors):
    """concat([xywh, conf, prob], -1) -> [B,S,S,3,5+C] (utils.py:128-131)."""
    xywh, conf, prob = decode(conv_output, scale, output_size, num_classes, strides, xyscale, anchors)
    return torch.cat([xywh, conf, prob], dim=-1)


def decode_predict(conv_output, scale, output_size, num_classes, strides, xyscale, anchors):
    """-> (pred_xywh [B,3S^2,4], pred_prob [B,3S^2,C] = conf*prob), candidates ordered
    (row, col, anchor) (utils.py:133-140)."""
    t = _as_5d(conv_output, num_classes)
    if int(output_size) != t.shape[1]:
        raise ValueError("output_size %s does not match the conv map (%d)" % (output_size, t.shape[1]))
    anc = _anchors_array(anchors)[scale]
    r = decode_heads([t], num_classes, [strides[scale]], [xyscale[scale]], anc[None])
    n = r["N"]
    return r["boxes"][:, :n], r["scores_cm"][:, :, :n].transpose(1, 2)


class Filtered(tuple):
    """(boxes, scores) as the reference returns them, plus .counts / .src_idx /
    .scores_cm (class-major storage) / .boxes_full for callers that stay on the GPU."""

    def __new__(cls, boxes, scores, **extra):
        self = super().__new__(cls, (boxes, scores))
        for k, v in extra.items():
            setattr(self, k, v)
        return self


def filter_bboxes(bboxes, scores, score_threshold, image_size):
    """Keep candidates with max_c score >= score_threshold, in order, and convert
    (cx,cy,w,h) pixels to (ymin,xmin,ymax,xmax)/image_size, unclipped (utils.py:142-167).

    bboxes [B,N,4], scores [B,N,C].  Every image is compacted on its own (the
    reference's flatten-and-reshape is only defined for B = 1); the returned tensors
    are padded with zeros to the largest survivor count, ``.counts`` holds the true
    counts and ``.src_idx`` the survivors' indices in the input.
    """
    _lib.require_gpu(bboxes, scores)
    if bboxes.dim() != 3 or scores.dim() != 3 or bboxes.shape[-1] != 4 or bboxes.shape[:2] != scores.shape[:2]:
        raise ValueError("expected bboxes [B,N,4] and scores [B,N,C]")
    B, N, C = scores.shape
    bx = bboxes.float().contiguous()
    sc = scores.float()
    cap = _round4(N)
    dev = scores.device
    boxes = torch.empty((B, cap, 4), device=dev, dtype=torch.float32)
    out_cm = torch.empty((B, C, cap), device=dev, dtype=torch.float32)
    src_idx = torch.empty((B, cap), device=dev, dtype=torch.int32)
    counts = torch.empty((B,), device=dev, dtype=torch.int32)
    L = _lib.lib()
    ws_bytes = L.ymx_filter_workspace_bytes(B, N)
    ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
    sb, sn, scs = sc.stride()
    _lib.launch("ymx_filter_boxes", boxes, _lib.ptr(bx), _lib.ptr(sc), sb, sn, scs, B, N, C, float(score_threshold),
                                  float(image_size), _lib.ptr(boxes), _lib.ptr(out_cm), _lib.ptr(src_idx),
                                  _lib.ptr(counts), cap, _lib.ptr(ws), ws_bytes)
    # one host read for both the largest survivor count and the kernel's status word
    host = torch.cat([counts.max().reshape(1), ws[:4].view(torch.int32)]).cpu()
    if int(host[1]) != 0:
        raise RuntimeError("yolo_amd: filter_bboxes: the compaction kernel timed out waiting for a "
                           "predecessor segment (device fault?); results are invalid")
    mmax = int(host[0])
    ar = torch.arange(mmax, device=dev)[None, :]
    valid = ar < counts[:, None]
    b_out = torch.where(valid[..., None], boxes[:, :mmax], torch.zeros((), device=dev))
    s_out = torch.where(valid[..., None], out_cm[:, :, :mmax].transpose(1, 2), torch.zeros((), device=dev))
    i_out = torch.where(valid, src_idx[:, :mmax], torch.full((), -1, device=dev, dtype=torch.int32))
    return Filtered(b_out, s_out, counts=counts, src_idx=i_out, scores_cm=out_cm, boxes_full=boxes,
                    src_idx_full=src_idx)


class Detections(tuple):
    """(boxes, scores, classes, valid_detections) plus .keep_idx."""

    def __new__(cls, boxes, scores, classes, valid, keep_idx):
        self = super().__new__(cls, (boxes, scores, classes, valid))
        self.keep_idx = keep_idx
        return self


def nms_class_major(boxes, scores_cm, counts, src_idx, max_output_size_per_class, max_total_size,
                    iou_threshold, score_threshold):
    """ymx_nms_combined on boxes [B,cap,4] and class-major scores [B,C,cap]."""
    _lib.require_gpu(boxes, scores_cm)
    B, C, cap = scores_cm.shape
    dev = boxes.device
    T = int(max_total_size)
    ob = torch.empty((B, T, 4), device=dev, dtype=torch.float32)
    os_ = torch.empty((B, T), device=dev, dtype=torch.float32)
    oc = torch.empty((B, T), device=dev, dtype=torch.float32)
    ov = torch.empty((B,), device=dev, dtype=torch.int32)
    ok = torch.empty((B, T), device=dev, dtype=torch.int32)
    L = _lib.lib()
    ws_bytes = L.ymx_nms_workspace_bytes(B, C, int(max_output_size_per_class))
    ws = torch.empty((max(ws_bytes, 16),), device=dev, dtype=torch.uint8)
    _lib.launch("ymx_nms_combined", boxes, _lib.ptr(boxes), _lib.ptr(scores_cm), _lib.ptr(counts), _lib.ptr(src_idx), B, cap,
                                  C, int(max_output_size_per_class), T, float(iou_threshold),
                                  float(score_threshold), _lib.ptr(ob), _lib.ptr(os_), _lib.ptr(oc), _lib.ptr(ov),
                                  _lib.ptr(ok), _lib.ptr(ws), ws_bytes)
    return Detections(ob, os_, oc, ov, ok)


def combined_non_max_suppression(boxes, scores, max_output_size_per_class, max_total_size, iou_threshold=0.5,
                                 score_threshold=float("-inf"), counts=None, src_idx=None):
    """The call the reference makes at utils.py:184-192, on GPU tensors.

    boxes [B,M,1,4] or [B,M,4] (ymin,xmin,ymax,xmax; shared by all classes), scores
    [B,M,C].  Candidates need score > score_threshold; within a class they are visited
    by descending score, ties by ascending index, and kept iff IoU <= iou_threshold with
    every kept box; the image's kept entries are ordered by (score desc, class asc,
    index asc), cut to max_total_size and clipped to [0,1].  counts [B] limits the
    valid candidates per image.  Returns (boxes [B,T,4], scores [B,T], classes [B,T]
    float32, valid_detections [B] int32) with .keep_idx [B,T] (-1 padding).
    """
    _lib.require_gpu(boxes, scores)
    if boxes.dim() == 4:
        if boxes.shape[2] != 1:
            raise ValueError("only class-shared boxes (q = 1) are supported")
        boxes = boxes[:, :, 0]
    B, M, C = scores.shape
    cap = _round4(M)
    if scores.stride(2) == scores.shape[1] and scores.stride(1) == 1 and M == cap and scores.dtype == torch.float32:
        cm = scores.transpose(1, 2)                      # already class-major in memory
    else:
        cm = torch.zeros((B, C, cap), device=scores.device, dtype=torch.float32)
        cm[:, :, :M] = scores.float().transpose(1, 2)
    bx = torch.zeros((B, cap, 4), device=boxes.device, dtype=torch.float32)
    bx[:, :M] = boxes.float()
    if counts is None:
        counts = torch.full((B,), M, device=boxes.device, dtype=torch.int32)
    if src_idx is not None and src_idx.shape[1] != cap:
        s2 = torch.full((B, cap), -1, device=boxes.device, dtype=torch.int32)
        s2[:, :src_idx.shape[1]] = src_idx
        src_idx = s2
    return nms_class_major(bx, cm.contiguous(), counts.to(torch.int32).contiguous(),
                           None if src_idx is None else src_idx.contiguous(),
                           max_output_size_per_class, max_total_size, iou_threshold, score_threshold)


def bbox_nms(model_output, image_size, max_outputs_per_class=32, max_outputs=32, iou_threshold=0.45,
             score_threshold=0.25, num_classes=None, strides=None, xyscale=None, anchors=None):
    """decode every scale -> filter -> combined NMS (utils.py:169-194).

    model_output = (sbbox, mbbox, lbbox) conv maps, each [B,S,S,3(5+C)] or [B,S,S,3,5+C].
    The reference's body calls decode_predict with 3 of its 7 arguments and concatenates
    the scales on the last axis, so it cannot run; this is the composition it intends:
    decode each scale with strides/xyscale/anchors from the package config, concatenate
    on the candidate axis, filter with >= score_threshold, then combined NMS with the
    same threshold.  Returns (boxes [B,T,4], scores [B,T], classes [B,T],
    valid_detections [B]); ``.keep_idx`` indexes the unfiltered candidate list.
    """
    heads = list(model_output)
    if num_classes is None:
        last = heads[0].shape[-1]
        num_classes = (last // 3 - 5) if heads[0].dim() == 4 else (last - 5)
    n = len(heads)
    strides = list(_YOLO["strides"] if strides is None else strides)[:n]
    xyscale = list(_YOLO["xyscale"] if xyscale is None else xyscale)[:n]
    anc = _anchors_array(_YOLO["anchors"] if anchors is None else anchors)[:n]
    r = decode_heads(heads, num_classes, strides, xyscale, anc, filter=True, score_threshold=score_threshold,
                     image_size=image_size)
    return nms_class_major(r["boxes"], r["scores_cm"], r["counts"], r["src_idx"], max_outputs_per_class,
                           max_outputs, iou_threshold, score_threshold)


def _iou_call(bboxes1, bboxes2, giou):
    _lib.require_gpu(bboxes1, bboxes2)
    b1, b2 = torch.broadcast_tensors(bboxes1.float(), bboxes2.float())
    if b1.shape[-1] != 4:
        raise ValueError("boxes must end in a (cx,cy,w,h) axis of 4")
    b1 = b1.contiguous()
    b2 = b2.contiguous()
    out = torch.empty(b1.shape[:-1], device=b1.device, dtype=torch.float32)
    _lib.launch("ymx_bbox_iou", out, _lib.ptr(b1), _lib.ptr(b2), _lib.ptr(out), out.numel(), 1 if giou else 0)
    return out


def bbox_iou(bboxes1, bboxes2):
    """IoU of (cx,cy,w,h) boxes with broadcasting; 0 where the union is 0 (utils.py:33-63)."""
    return _iou_call(bboxes1, bboxes2, False)


def bbox_giou(bboxes1, bboxes2):
    """Generalised IoU of (cx,cy,w,h) boxes (utils.py:65-105)."""
    return _iou_call(bboxes1, bboxes2, True)


def image_preprocess(image, bboxes, train_input_size, literal=False, dtype=torch.float32):
    """Letterbox resize of an image (or a batch of equally sized images) and of its boxes
    (utils.py:5-31).

    image [H,W,3] or [B,H,W,3], uint8 or float32, on the GPU; bboxes [n, >=4] (xmin, ymin,
    xmax, ymax, ...) numpy array or None.  Returns (image [T,T,3] or [B,T,T,3] in ``dtype``
    = value / 256 with 128-filled borders, boxes scaled and shifted as utils.py:29-30).
    literal=True keeps the reference's cv2.resize argument order (utils.py:21), which swaps
    the resized height and width of a non-square image; the default resizes to
    (new_height rows, new_width columns), what the box arithmetic assumes."""
    _lib.require_gpu(image)
    single = image.dim() == 3
    x = image.unsqueeze(0) if single else image
    if x.dim() != 4 or x.shape[-1] != 3:
        raise ValueError("image: expected [H,W,3] or [B,H,W,3], got %s" % (tuple(image.shape),))
    if x.dtype not in (torch.uint8, torch.float32):
        raise ValueError("image must be uint8 or float32")
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("output dtype must be float32 or bfloat16")
    x = x.contiguous()
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    T = int(train_input_size)
    out = torch.empty((B, T, T, 3), device=x.device, dtype=dtype)
    _lib.launch("ymx_letterbox", x, _lib.ptr(x), 2 if x.dtype == torch.uint8 else 0, B, H, W, _lib.ptr(out),
                _lib.dtype_code(dtype), T, 1 if literal else 0)
    boxes = None
    if bboxes is not None:
        scale = min(T / W, T / H)
        new_h, new_w = int(scale * H), int(scale * W)
        pad_h, pad_w = (T - new_h) // 2, (T - new_w) // 2
        boxes = np.array(bboxes, np.float64, copy=True)
        boxes[:, [0, 2]] = boxes[:, [0, 2]] * scale + pad_w
        boxes[:, [1, 3]] = boxes[:, [1, 3]] * scale + pad_h
    return (out[0] if single else out), boxes


def class_colors(num_classes):
    """The colour table of draw_bbox (utils.py:211-213): hue = class / num_classes at full
    saturation and value, as 8-bit RGB, [num_classes, 3] uint8."""
    import colorsys
    cols = [colorsys.hsv_to_rgb(1.0 * c / num_classes, 1.0, 1.0) for c in range(num_classes)]
    return np.array([[int(r * 255), int(g * 255), int(b * 255)] for r, g, b in cols], np.uint8)


LABEL_BYTES = 48          # characters kept per label; longer strings are cut


def label_text(name, score):
    """The label string of utils.py:234."""
    return '%s: %.2f' % (name, float(score))


def encode_labels(labels, max_boxes, label_bytes=LABEL_BYTES):
    """labels: per image a sequence of strings (one per detection, '' or None = no label) ->
    uint8 array [B, max_boxes, label_bytes], zero-padded, characters outside Latin-1 replaced
    by code 127 (drawn as a hollow box)."""
    out = np.zeros((len(labels), max_boxes, label_bytes), np.uint8)
    for b, row in enumerate(labels):
        if len(row) > max_boxes:
            raise ValueError("image %d has %d labels for %d detection slots" % (b, len(row), max_boxes))
        for j, text in enumerate(row):
            codes = [ord(c) if 0 < ord(c) < 256 else 127 for c in (text or "")][:label_bytes]
            out[b, j, :len(codes)] = codes
    return out


def draw_boxes(images, boxes, classes, valid_detections, num_classes, thickness=None, labels=None, text_scale=2):
    """Outlines detections on a batch of images in place (the cv2.rectangle calls of
    utils.py:216-231).  images [B,H,W,3] uint8 on the GPU, the other arguments as
    bbox_nms returns them.  thickness None = int(0.6 * (H + W) / 600), at least 1.
    labels: None, or per image a sequence of strings (one per detection) drawn as the
    show_label branch does (utils.py:233-240): a filled rectangle in the class colour above
    the box corner and the text in black, in the library's 5x7 bitmap typeface with
    text_scale image pixels per font pixel."""
    _lib.require_gpu(images, boxes, classes, valid_detections)
    if images.dim() != 4 or images.shape[-1] != 3 or images.dtype != torch.uint8 or not images.is_contiguous():
        raise ValueError("images: expected contiguous uint8 [B,H,W,3]")
    B, H, W = images.shape[0], images.shape[1], images.shape[2]
    bx = boxes.float().contiguous()
    cl = classes.float().contiguous()
    vd = valid_detections.to(torch.int32).contiguous()
    if bx.dim() != 3 or bx.shape[0] != B or bx.shape[2] != 4 or cl.shape != bx.shape[:2] or vd.shape != (B,):
        raise ValueError("boxes [B,M,4], classes [B,M] and valid_detections [B] are required")
    if thickness is None:
        thickness = int(0.6 * (H + W) / 600)
    colors = torch.from_numpy(class_colors(num_classes)).to(images.device)
    if labels is None:
        _lib.launch("ymx_draw_boxes", images, _lib.ptr(images), B, H, W, _lib.ptr(bx), _lib.ptr(cl), _lib.ptr(vd),
                    bx.shape[1], _lib.ptr(colors), int(num_classes), max(int(thickness), 1))
        return images
    if len(labels) != B:
        raise ValueError("labels: one sequence of strings per image (%d given for %d images)" % (len(labels), B))
    lab = torch.from_numpy(encode_labels(labels, bx.shape[1])).to(images.device)
    _lib.launch("ymx_draw_boxes_labeled", images, _lib.ptr(images), B, H, W, _lib.ptr(bx), _lib.ptr(cl), _lib.ptr(vd),
                bx.shape[1], _lib.ptr(colors), int(num_classes), max(int(thickness), 1), _lib.ptr(lab), lab.shape[2],
                int(text_scale))
    return images


def draw_bbox(images, model_output, image_size, num_classes, classes=None, show_label=True, text_scale=2):
    """NMS on the model output, then every detection outlined on its image and, with
    show_label, captioned '<class name>: <score with two decimals>' (utils.py:197-242).

    images [B,H,W,3] uint8 on the GPU (drawn on in place, as the reference draws on its
    arrays); returns the list of the B images.  classes: the class names (default: the
    class indices as strings, utils.py:206-207).  The captions are formatted on the host, so
    show_label=True reads the detections back (one synchronisation); the typeface is the
    library's bitmap font, not OpenCV's."""
    det = bbox_nms(model_output, image_size, num_classes=num_classes)
    labels = None
    if show_label:
        names = list(classes) if classes else [str(i) for i in range(num_classes)]
        scores, cls, valid = det[1].float().cpu().numpy(), det[2].float().cpu().numpy(), det[3].cpu().numpy()
        labels = []
        for b in range(images.shape[0]):
            row = []
            for j in range(min(int(valid[b]), cls.shape[1])):
                c = int(cls[b, j])
                row.append(label_text(names[c], scores[b, j]) if 0 <= c < min(num_classes, len(names)) else "")
            labels.append(row)
    draw_boxes(images, det[0], det[2], det[3], num_classes, labels=labels, text_scale=text_scale)
    return [images[i] for i in range(images.shape[0])]
