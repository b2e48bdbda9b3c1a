"""numpy fp32 restatement of the reference's YOLO loss.

Test infrastructure only (see ``oracle/__init__.py``); parity unpinned: the reference holds
no fixture for the loss and TensorFlow cannot run here, so this file restates
/root/reference/losses.py from its text (tf.nn.sigmoid_cross_entropy_with_logits by its
published definition) and is pinned by the scalar known-answer computation in
tests/test_oracle.py.

Follows
  yolo_loss                 losses.py:10-46
  get_yolo_loss_at_scales   losses.py:48-70 (closure body 52-66)
"""
import numpy as np

from . import postproc_np as P

F32 = np.float32


def sigmoid_cross_entropy_with_logits(labels, logits):
    """max(x, 0) - x*z + log(1 + exp(-|x|)) (TensorFlow's documented stable form)."""
    x = np.asarray(logits, F32)
    z = np.asarray(labels, F32)
    return (np.maximum(x, F32(0)) - x * z + np.log1p(np.exp(-np.abs(x)))).astype(F32)


def yolo_loss(pred, conv, label, bboxes, input_size, iou_loss_threshold):
    """(giou_loss, conf_loss, prob_loss) scalars; losses.py:10-46.

    pred = decode_train(conv) [B,S,S,3,5+C]; conv raw map; label same shape (xywh, respond,
    class probabilities); bboxes [B,NB,4] centre-format ground truth.
    """
    pred = np.asarray(pred, F32); conv = np.asarray(conv, F32)
    label = np.asarray(label, F32); bboxes = np.asarray(bboxes, F32)
    conv_raw_conf = conv[..., 4:5]
    conv_raw_prob = conv[..., 5:]
    pred_xywh = pred[..., 0:4]
    pred_conf = pred[..., 4:5]
    label_xywh = label[..., 0:4]
    respond_bbox = label[..., 4:5]
    label_prob = label[..., 5:]

    giou = P.bbox_giou(pred_xywh, label_xywh)[..., None]                       # :21
    input_size = F32(input_size)
    bbox_loss_scale = F32(2.0) - F32(1.0) * label_xywh[..., 2:3] * label_xywh[..., 3:4] / (input_size ** 2)   # :24
    giou_loss = respond_bbox * bbox_loss_scale * (F32(1) - giou)              # :25

    iou = P.bbox_iou(pred_xywh[:, :, :, :, None, :], bboxes[:, None, None, None, :, :])   # :27
    max_iou = iou.max(axis=-1)[..., None]                                      # :28
    respond_bgd = (F32(1.0) - respond_bbox) * (max_iou < F32(iou_loss_threshold)).astype(F32)   # :30
    conf_focal = (respond_bbox - pred_conf) ** 2                               # :32
    bce = sigmoid_cross_entropy_with_logits(respond_bbox, conv_raw_conf)
    conf_loss = conf_focal * (respond_bbox * bce + respond_bgd * bce)          # :34-38
    prob_loss = respond_bbox * sigmoid_cross_entropy_with_logits(label_prob, conv_raw_prob)    # :40

    def reduce(t):                                                             # :42-44
        return F32(np.mean(np.sum(t.astype(F32), axis=(1, 2, 3, 4), dtype=F32), dtype=F32))

    return reduce(giou_loss), reduce(conf_loss), reduce(prob_loss)



def loss_at_scale(target, output, scale, num_classes, strides, iou_loss_threshold, xyscale, anchors):
    """Body of one closure of get_yolo_loss_at_scales for a given ``scale``; losses.py:52-66.

    target [B, S*S*3*(5+C) + 128*4], output [B,S,S,3*(5+C)] -> (total, (giou, conf, prob)).
    """
    output = np.asarray(output, F32)
    target = np.asarray(target, F32)
    B, S = output.shape[0], output.shape[1]
    input_size = S * strides[scale]
    shape = (B, S, S, 3, 5 + num_classes)
    split = int(np.prod(shape[1:]))
    out5 = output.reshape(shape)
    labels = target[:, :split].reshape(shape)
    bboxes = target[:, split:].reshape(B, 128, 4)
    decoded = P.decode_train(out5, scale, S, num_classes, strides, xyscale, anchors)
    g, c, p = yolo_loss(decoded, out5, labels, bboxes, input_size, iou_loss_threshold)
    return F32(g + c + p), (g, c, p)


def get_yolo_loss_at_scales(scales, num_classes, strides, iou_loss_threshold, xyscale, anchors, late_binding=True):
    """List of loss(target, output) callables in the order of scales[::-1]; losses.py:48-70.

    The reference's closures capture the loop variable ``scale`` and are traced only when
    first called, after the loop has ended, so EVERY closure uses the last value the loop
    assigned, scales[::-1][-1] = scales[0] (late_binding=True reproduces that);
    late_binding=False gives each closure the scale of its own iteration.
    """
    order = list(scales)[::-1]
    losses = []
    for sc in order:
        use = list(scales)[0] if late_binding else sc
        losses.append(lambda target, output, _s=use: loss_at_scale(target, output, _s, num_classes, strides,
                                                                   iou_loss_threshold, xyscale, anchors)[0])
    return losses
