"""YOLO loss (forward), same names and signatures as the reference's losses.py.

Mirrors /root/reference/losses.py:10-70.  One fused HIP launch per scale (ymx_yolo_loss)
reads the raw head map, the label tensor and the ground-truth boxes once and produces the
three loss terms; the [B,S,S,3,128] IoU tensor of losses.py:27 is never materialised.
The closures are differentiable with respect to the head output: a second fused launch
(ymx_yolo_loss_grad) produces d loss / d output, wired into torch.autograd, so
``loss(target, output).backward()`` works on the GPU.  Gradients with respect to the
targets are not defined (as in the reference, where they are data).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .utils import _anchors_array

N_BOXES = 128          # ground-truth slots per image in the target layout (losses.py:62)


def _launch(conv5, conv_ld, pred5, label5, label_stride, bboxes, bbox_stride, stride, xyscale, anchors3x2, iou_loss_threshold, per_image=False):
    B, S, C = conv5.shape[0], conv5.shape[1], conv5.shape[4] - 5
    dev = conv5.device
    out = torch.empty((3,), device=dev, dtype=torch.float32)
    per = torch.empty((B, 3), device=dev, dtype=torch.float32) if per_image else None
    L = _lib.lib()
    ws_bytes = L.ymx_yolo_loss_workspace_bytes(B, S)
    ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
    anc = (ctypes.c_float * 6)(*[float(v) for v in np.asarray(anchors3x2, np.float32).reshape(-1)[:6]])
    _lib.launch("ymx_yolo_loss", out, _lib.ptr(conv5), int(conv_ld), _lib.ptr(pred5), _lib.ptr(label5), int(label_stride),
                _lib.ptr(bboxes), int(bbox_stride), B, S, C, int(bboxes.shape[1]), float(stride), float(xyscale), anc, float(iou_loss_threshold),
                _lib.ptr(out), _lib.ptr(per), _lib.ptr(ws), ws_bytes)
    return out, per


def _check5(t, name):
    _lib.require_gpu(t)
    if t.dim() != 5 or t.shape[3] != 3 or t.shape[1] != t.shape[2]:
        raise ValueError("%s: expected [B,S,S,3,5+C], got %s" % (name, tuple(t.shape)))
    return t.float().contiguous()


def yolo_loss(pred, conv, label, bboxes, input_size, iou_loss_threshold):
    """(giou_loss, conf_loss, prob_loss), each a 0-dim float32 GPU tensor (losses.py:10-46).

    pred [B,S,S,3,5+C] decoded (xywh, conf, ...), conv the raw map, label (xywh, respond,
    class probabilities), bboxes [B,NB,4] centre-format ground truth.
    """
    conv5, pred5, label5 = _check5(conv, "conv"), _check5(pred, "pred"), _check5(label, "label")
    if pred5.shape != conv5.shape or label5.shape != conv5.shape:
        raise ValueError("pred, conv and label must have the same shape")
    bx = bboxes.float().contiguous()
    if bx.dim() != 3 or bx.shape[0] != conv5.shape[0] or bx.shape[2] != 4:
        raise ValueError("bboxes: expected [B,NB,4], got %s" % (tuple(bx.shape),))
    S = conv5.shape[1]
    out, _ = _launch(conv5, 0, pred5, label5, 0, bx, 0, float(input_size) / S, 1.0, np.ones((3, 2), np.float32),
                     iou_loss_threshold)
    return out[0], out[1], out[2]


def _split_target(target, output, num_classes):
    if output.dim() != 4 or output.shape[1] != output.shape[2] or output.shape[3] != 3 * (5 + num_classes):
        raise ValueError("output: expected [B,S,S,3*(5+C)], got %s" % (tuple(output.shape),))
    B, S = output.shape[0], output.shape[1]
    D = 5 + num_classes
    split = S * S * 3 * D
    if target.dim() != 2 or target.shape[0] != B or target.shape[1] != split + N_BOXES * 4:
        raise ValueError("target: expected [B, %d], got %s" % (split + N_BOXES * 4, tuple(target.shape)))
    tgt = target.detach().float().contiguous()
    # labels and boxes are read in place from the packed target rows (image stride = row length)
    return tgt, tgt[:, :split], tgt[:, split:].unflatten(1, (N_BOXES, 4))


class _LossAtScale(torch.autograd.Function):
    """terms [3] = (giou, conf, prob) of one scale as a function of the head output."""

    @staticmethod
    def forward(ctx, output, target, scale, num_classes, strides, iou_loss_threshold, xyscale, anchors):
        tgt, labels, bboxes = _split_target(target, output, num_classes)
        B, S = output.shape[0], output.shape[1]
        out5 = output.detach().float().contiguous().view(B, S, S, 3, 5 + num_classes)
        anc = _anchors_array(anchors)[scale]
        terms, _ = _launch(out5, 0, None, labels, tgt.shape[1], bboxes, tgt.shape[1], strides[scale], xyscale[scale], anc,
                           iou_loss_threshold)
        ctx.save_for_backward(out5, tgt)
        ctx.cfg = (num_classes, float(strides[scale]), float(xyscale[scale]), anc, float(iou_loss_threshold), output.dtype)
        return terms

    @staticmethod
    def backward(ctx, grad_terms):
        out5, tgt = ctx.saved_tensors
        C, stride, xys, anc, thr, dtype = ctx.cfg
        B, S = out5.shape[0], out5.shape[1]
        split = S * S * 3 * (5 + C)
        wts = grad_terms.detach().float().contiguous()                  # upstream gradient of the three terms, stays on the GPU
        ancf = (ctypes.c_float * 6)(*[float(v) for v in np.asarray(anc, np.float32).reshape(-1)[:6]])
        dconv = torch.empty_like(out5)
        labels, bboxes = tgt[:, :split], tgt[:, split:]
        _lib.launch("ymx_yolo_loss_grad", dconv, _lib.ptr(out5), 0, _lib.ptr(labels), int(tgt.shape[1]), _lib.ptr(bboxes),
                    int(tgt.shape[1]), B, S, C, N_BOXES, stride, xys, ancf, thr, _lib.ptr(wts), _lib.ptr(dconv))
        return dconv.view(B, S, S, 3 * (5 + C)).to(dtype), None, None, None, None, None, None, None


def loss_at_scale(target, output, scale, num_classes, strides, iou_loss_threshold, xyscale, anchors, terms=False):
    """Body of one closure of get_yolo_loss_at_scales (losses.py:52-66) for ``scale``:
    target [B, S*S*3*(5+C) + 128*4], output [B,S,S,3*(5+C)] -> giou + conf + prob (0-dim
    tensor); with terms=True the three terms as a [3] tensor.  The decode of ``output``
    (decode_train) happens inside the kernel.  Differentiable with respect to ``output``."""
    _lib.require_gpu(target, output)
    out = _LossAtScale.apply(output, target, scale, num_classes, strides, iou_loss_threshold, xyscale, anchors)
    return out if terms else out.sum()


def get_yolo_loss_at_scales(scales, num_classes, strides, iou_loss_threshold, xyscale, anchors, late_binding=False):
    """List of loss(target, output) callables, one per scale, in the order of scales[::-1]
    (losses.py:48-70).

    In the reference every closure reads the loop variable ``scale`` when it is first traced,
    after the loop has ended, so all of them use scales[0]; late_binding=True reproduces
    that, the default gives each closure the scale of its own iteration (what the code
    evidently intends)."""
    order = list(scales)[::-1]
    losses = []
    for sc in order:
        use = list(scales)[0] if late_binding else sc

        def loss(target, output, _scale=use):
            return loss_at_scale(target, output, _scale, num_classes, strides, iou_loss_threshold, xyscale, anchors)

        losses.append(loss)
    return losses
