"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image utilities of the reference
(parity unpinned: OpenCV is not installed here, so cv2.resize / cv2.rectangle themselves
cannot be run; the reference holds no image fixtures).

  letterbox    /root/reference/utils.py:16-26 (image half of image_preprocess); the resize
               follows the published algorithm of cv2.resize(..., interpolation=INTER_LINEAR)
               for float images: half-pixel centres, edge replication, rows then columns
  draw_boxes   /root/reference/utils.py:216-231 (the cv2.rectangle calls of draw_bbox) with
               this build's definition of an outline of thickness t
"""
import colorsys

import numpy as np


def _coords(n_dst, n_src):
    """Per destination index: (i0, i1, f) of cv2's linear resize."""
    ratio = np.float64(n_src) / np.float64(n_dst)
    v = ((np.arange(n_dst, dtype=np.float64) + 0.5) * ratio - 0.5).astype(np.float32)
    i = np.floor(v).astype(np.int64)
    f = (v - i.astype(np.float32)).astype(np.float32)
    low = i < 0
    i[low] = 0
    f[low] = 0
    high = i >= n_src - 1
    i[high] = n_src - 1
    f[high] = 0
    return i, np.minimum(i + 1, n_src - 1), f


def resize_linear(img, rows, cols):
    """img [H,W,3] (any real dtype) -> [rows, cols, 3] float32."""
    s = np.asarray(img).astype(np.float32)
    y0, y1, fy = _coords(rows, s.shape[0])
    x0, x1, fx = _coords(cols, s.shape[1])
    ax = (np.float32(1) - fx)[None, :, None]
    bx = fx[None, :, None]
    top = (s[y0][:, x0] * ax).astype(np.float32) + (s[y0][:, x1] * bx).astype(np.float32)
    bot = (s[y1][:, x0] * ax).astype(np.float32) + (s[y1][:, x1] * bx).astype(np.float32)
    ay = (np.float32(1) - fy)[:, None, None]
    by = fy[:, None, None]
    return ((top * ay).astype(np.float32) + (bot * by).astype(np.float32)).astype(np.float32)


def letterbox(image, target, literal=False):
    """utils.py:16-26 for one image [H,W,3] -> [target,target,3] float32."""
    h, w = image.shape[0], image.shape[1]
    scale = min(target / w, target / h)
    new_h, new_w = int(scale * h), int(scale * w)
    pad_h, pad_w = (target - new_h) // 2, (target - new_w) // 2
    out = np.full((target, target, 3), 128.0, np.float32)
    if literal:      # cv2.resize(image, (new_height, new_width)): dsize is (width, height)
        out[pad_w:pad_w + new_w, pad_h:pad_h + new_h] = resize_linear(image, new_w, new_h)
    else:
        out[pad_h:pad_h + new_h, pad_w:pad_w + new_w] = resize_linear(image, new_h, new_w)
    return (out / np.float32(256)).astype(np.float32)


def class_colors(num_classes):
    cols = [colorsys.hsv_to_rgb(1.0 * c / num_classes, 1.0, 1.0) for c in range(num_classes)]
    return np.array([[int(r * 255), int(g * 255), int(b * 255)] for r, g, b in cols], np.uint8)


def draw_boxes(image, boxes, classes, n_valid, num_classes, thickness):
    """Outlines on one image [H,W,3] uint8, boxes [M,4] normalised (ymin,xmin,ymax,xmax),
    drawn one after another (utils.py:217-231)."""
    img = image.copy()
    h, w = img.shape[0], img.shape[1]
    t = max(int(thickness), 1)
    lo, hi = (t - 1) // 2, t // 2
    colors = class_colors(num_classes)
    for j in range(int(n_valid)):
        cls = int(classes[j])
        if cls < 0 or cls >= num_classes:
            continue
        y1 = int(np.float32(boxes[j][0]) * np.float32(h)); x1 = int(np.float32(boxes[j][1]) * np.float32(w))
        y2 = int(np.float32(boxes[j][2]) * np.float32(h)); x2 = int(np.float32(boxes[j][3]) * np.float32(w))
        ya, yb, xa, xb = min(y1, y2), max(y1, y2), min(x1, x2), max(x1, x2)
        for y in range(max(ya - lo, 0), min(yb + hi, h - 1) + 1):
            for x in range(max(xa - lo, 0), min(xb + hi, w - 1) + 1):
                on_v = (xa - lo <= x <= xa + hi) or (xb - lo <= x <= xb + hi)
                on_h = (ya - lo <= y <= ya + hi) or (yb - lo <= y <= yb + hi)
                if on_v or on_h:
                    img[y, x] = colors[cls]
    return img
