"""TEST INFRASTRUCTURE ONLY -- the reference's YOLO loss restated in torch (CPU, float64) so
that automatic differentiation gives the gradient of the loss with respect to the raw head
map, the quantity a training step back-propagates (parity unpinned, as every oracle here:
TensorFlow cannot run in this image).

Follows /root/reference/losses.py:10-46 (yolo_loss), :52-66 (closure body) and
/root/reference/utils.py:33-131 (bbox_iou, bbox_giou, decode, decode_train) line by line;
the forward value is checked against the numpy restatement (oracle/loss_np.py) in
tests/test_oracle.py, which ties the two together.  The background mask
``max_iou < threshold`` is a cast of a comparison and carries no gradient, in TensorFlow as
here.  Where two arguments of a min / max are exactly equal TensorFlow routes the gradient
to the first and torch splits it; the tests use inputs without such ties.
"""
import numpy as np
import torch

F64 = torch.float64


def _div_no_nan(x, y):
    safe = torch.where(y == 0, torch.ones_like(y), y)
    return torch.where(y == 0, torch.zeros_like(x), x / safe)


def _corners(b):
    return torch.cat([b[..., :2] - b[..., 2:] * 0.5, b[..., :2] + b[..., 2:] * 0.5], dim=-1)


def bbox_iou(b1, b2):                                                    # utils.py:33-63
    a1 = b1[..., 2] * b1[..., 3]
    a2 = b2[..., 2] * b2[..., 3]
    c1, c2 = _corners(b1), _corners(b2)
    lu = torch.maximum(c1[..., :2], c2[..., :2])
    rd = torch.minimum(c1[..., 2:], c2[..., 2:])
    sec = torch.clamp(rd - lu, min=0.0)
    inter = sec[..., 0] * sec[..., 1]
    return _div_no_nan(inter, a1 + a2 - inter)


def bbox_giou(b1, b2):                                                   # utils.py:65-105
    a1 = b1[..., 2] * b1[..., 3]
    a2 = b2[..., 2] * b2[..., 3]
    c1, c2 = _corners(b1), _corners(b2)
    lu = torch.maximum(c1[..., :2], c2[..., :2])
    rd = torch.minimum(c1[..., 2:], c2[..., 2:])
    sec = torch.clamp(rd - lu, min=0.0)
    inter = sec[..., 0] * sec[..., 1]
    union = a1 + a2 - inter
    iou = _div_no_nan(inter, union)
    elu = torch.minimum(c1[..., :2], c2[..., :2])
    erd = torch.maximum(c1[..., 2:], c2[..., 2:])
    es = erd - elu
    earea = es[..., 0] * es[..., 1]
    return iou - _div_no_nan(earea - union, earea)


def decode_train(conv5, scale, S, strides, xyscale, anchors):          # utils.py:107-131
    ys, xs = torch.meshgrid(torch.arange(S, dtype=F64), torch.arange(S, dtype=F64), indexing="ij")
    grid = torch.stack([xs, ys], dim=-1)[None, :, :, None, :]              # grid[row, col] = (x = col, y = row)
    xy = (torch.sigmoid(conv5[..., 0:2]) * xyscale[scale] - 0.5 * (xyscale[scale] - 1) + grid) * strides[scale]
    anc = torch.as_tensor(np.asarray(anchors, np.float64).reshape(3, 3, 2)[scale], dtype=F64)
    wh = torch.exp(conv5[..., 2:4]) * anc
    return torch.cat([xy, wh, torch.sigmoid(conv5[..., 4:5]), torch.sigmoid(conv5[..., 5:])], dim=-1)


def _sce(labels, logits):
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))


def yolo_loss(pred, conv, label, bboxes, input_size, iou_loss_threshold):   # losses.py:10-46
    respond = label[..., 4:5]
    giou = bbox_giou(pred[..., 0:4], label[..., 0:4])[..., None]
    scale = 2.0 - 1.0 * label[..., 2:3] * label[..., 3:4] / (float(input_size) ** 2)
    giou_loss = respond * scale * (1 - giou)
    iou = bbox_iou(pred[..., 0:4][:, :, :, :, None, :], bboxes[:, None, None, None, :, :])
    max_iou = iou.max(dim=-1).values[..., None]
    bgd = (1.0 - respond) * (max_iou < iou_loss_threshold).to(F64)
    focal = (respond - pred[..., 4:5]) ** 2
    bce = _sce(respond, conv[..., 4:5])
    conf_loss = focal * (respond * bce + bgd * bce)
    prob_loss = respond * _sce(label[..., 5:], conv[..., 5:])
    red = lambda t: t.sum(dim=(1, 2, 3, 4)).mean()
    return red(giou_loss), red(conf_loss), red(prob_loss)


def l2_penalty(kernels, coefficient=0.0005):
    """What Keras adds to the loss it minimises for the reference's convolutions:
    kernel_regularizer=tf.keras.regularizers.l2(0.0005) on every Conv2D kernel, biases left out
    (/root/reference/layers.py:59-63); l2(c) is c * sum(w ** 2)."""
    total = torch.zeros((), dtype=F64)
    for k in kernels:
        total = total + coefficient * k.to(F64).square().sum()
    return total


def loss_and_grad(target, output, scale, num_classes, strides, iou_loss_threshold, xyscale, anchors, weights=(1.0, 1.0, 1.0)):
    """((giou, conf, prob) as floats, d(w0*giou + w1*conf + w2*prob)/d output as a float64
    array shaped like ``output`` [B,S,S,3*(5+C)])."""
    out = torch.as_tensor(np.asarray(output, np.float64)).clone().requires_grad_(True)
    tgt = torch.as_tensor(np.asarray(target, np.float64))
    B, S = out.shape[0], out.shape[1]
    D = 5 + num_classes
    split = S * S * 3 * D
    conv5 = out.reshape(B, S, S, 3, D)
    label = tgt[:, :split].reshape(B, S, S, 3, D)
    bboxes = tgt[:, split:].reshape(B, 128, 4)
    pred = decode_train(conv5, scale, S, strides, xyscale, anchors)
    g, c, p = yolo_loss(pred, conv5, label, bboxes, S * strides[scale], iou_loss_threshold)
    (weights[0] * g + weights[1] * c + weights[2] * p).backward()
    return (float(g.detach()), float(c.detach()), float(p.detach())), out.grad.numpy()
