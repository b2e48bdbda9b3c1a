"""ctypes front end of the plain-C restatement in ``oracle/c/postproc_ref.c``.

Test infrastructure only (see ``oracle/__init__.py``); parity unpinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "c", "postproc_ref.c")
_SO = os.path.join(_HERE, "c", "libyolo_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C restatement with gcc (strict fp32, no contraction)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                               "-shared", "-o", _SO, _SRC, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.yo_filter.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_i)


def decode_all(heads, num_classes, strides=(8, 16, 32), xyscale=(1.2, 1.1, 1.05), anchors=None):
    from .postproc_np import ANCHORS
    anchors = np.ascontiguousarray(ANCHORS if anchors is None else anchors, np.float32).reshape(3, 3, 2)
    boxes, scores = [], []
    for k, h in enumerate(heads):
        h = np.ascontiguousarray(h, np.float32)
        B, S = h.shape[0], h.shape[1]
        xywh = np.empty((B, 3 * S * S, 4), np.float32)
        sc = np.empty((B, 3 * S * S, num_classes), np.float32)
        anc = np.ascontiguousarray(anchors[k])
        lib().yo_decode_scale(_fp(h), B, S, num_classes, ctypes.c_float(strides[k]),
                              ctypes.c_float(xyscale[k]), _fp(anc), _fp(xywh), _fp(sc))
        boxes.append(xywh)
        scores.append(sc)
    return np.concatenate(boxes, 1), np.concatenate(scores, 1)


def filter_bboxes(xywh, scores, thr, image_size):
    xywh = np.ascontiguousarray(xywh, np.float32)
    scores = np.ascontiguousarray(scores, np.float32)
    B, N, C = scores.shape
    boxes = np.empty((B, N, 4), np.float32)
    sout = np.empty((B, N, C), np.float32)
    counts = np.empty((B,), np.int32)
    src = np.empty((B, N), np.int32)
    mmax = lib().yo_filter(_fp(xywh), _fp(scores), B, N, C, ctypes.c_float(thr),
                           ctypes.c_float(image_size), _fp(boxes), _fp(sout), _ip(counts),
                           _ip(src), N)
    assert mmax >= 0
    return boxes[:, :mmax].copy(), sout[:, :mmax].copy(), counts, src[:, :mmax].copy()


def combined_nms(boxes, scores, counts=None, max_output_size_per_class=32, max_total_size=32,
                 iou_threshold=0.45, score_threshold=0.25, src_idx=None):
    boxes = np.ascontiguousarray(boxes, np.float32)
    scores = np.ascontiguousarray(scores, np.float32)
    B, M, C = scores.shape
    T = int(max_total_size)
    if counts is not None:
        counts = np.ascontiguousarray(counts, np.int32)
    if src_idx is not None:
        src_idx = np.ascontiguousarray(src_idx, np.int32)
        assert src_idx.shape == (B, M)
    ob = np.empty((B, T, 4), np.float32)
    os_ = np.empty((B, T), np.float32)
    oc = np.empty((B, T), np.float32)
    ov = np.empty((B,), np.int32)
    ok = np.empty((B, T), np.int32)
    lib().yo_combined_nms(_fp(boxes), _fp(scores), _ip(counts), B, M, C,
                          int(max_output_size_per_class), T, ctypes.c_float(iou_threshold),
                          ctypes.c_float(score_threshold), _ip(src_idx), _fp(ob), _fp(os_),
                          _fp(oc), _ip(ov), _ip(ok))
    return ob, os_, oc, ov, ok


def bbox_nms(heads, image_size, num_classes, max_outputs_per_class=32, max_outputs=32,
             iou_threshold=0.45, score_threshold=0.25):
    xywh, sc = decode_all(heads, num_classes)
    fb, fs, counts, src = filter_bboxes(xywh, sc, score_threshold, image_size)
    return combined_nms(fb, fs, counts, max_outputs_per_class, max_outputs, iou_threshold,
                        score_threshold, src_idx=src)
