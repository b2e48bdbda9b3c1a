"""Regenerates the golden fixtures in this directory from the CPU oracle.

The reference ships no golden vectors and cannot be imported here (TensorFlow is not
installed), so these fixtures pin *this build's* oracle (oracle/postproc_np.py,
oracle/net_torch.py), not TensorFlow's results: parity unpinned.  They guard the oracle
against silent change and give the GPU tests inputs/outputs that do not depend on the
random generators of the machine that runs them.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import postproc_np as P      # noqa: E402
from oracle import net_torch as N        # noqa: E402


def postproc_case(name, B, sides, C, seed, conf_bias, thr, per_class, total, iou):
    rng = np.random.default_rng(seed)
    heads = []
    for S in sides:
        h = rng.normal(0.0, 2.0, (B, S, S, 3, 5 + C)).astype(np.float32)
        h[..., 4] += conf_bias
        heads.append(h)
    image_size = sides[0] * 8
    xywh, sc = P.decode_all(heads, C)
    fb, fs, counts, src = P.filter_bboxes(xywh, sc, thr, image_size)
    ob, os_, oc, ov, ok = P.combined_nms(fb, fs, counts, per_class, total, iou, thr, src_idx=src)
    out = {"thr": np.float32(thr), "iou": np.float32(iou), "per_class": np.int32(per_class),
           "total": np.int32(total), "image_size": np.float32(image_size), "C": np.int32(C),
           "xywh": xywh, "scores": sc, "f_boxes": fb, "f_scores": fs, "f_counts": counts, "f_src": src,
           "n_boxes": ob, "n_scores": os_, "n_classes": oc, "n_valid": ov, "n_keep": ok}
    for i, h in enumerate(heads):
        out["head%d" % i] = h
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def nms_case(name, B, M, C, seed, thr, per_class, total, iou, quant=None):
    """Pre-decoded boxes: centres U[0,1), sides LogNormal, scores U^4 (many ties if quantised)."""
    rng = np.random.default_rng(seed)
    cy, cx = rng.random((B, M)), rng.random((B, M))
    h = np.exp(rng.normal(-2.0, 0.5, (B, M)))
    w = np.exp(rng.normal(-2.0, 0.5, (B, M)))
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
    scores = (rng.random((B, M, C)) ** 4).astype(np.float32)
    if quant:
        scores = (np.round(scores * quant) / quant).astype(np.float32)
    counts = np.array([M - 3 * b for b in range(B)], np.int32)
    ob, os_, oc, ov, ok = P.combined_nms(boxes, scores, counts, per_class, total, iou, thr)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), boxes=boxes, scores=scores, counts=counts,
                        thr=np.float32(thr), iou=np.float32(iou), per_class=np.int32(per_class),
                        total=np.int32(total), n_boxes=ob, n_scores=os_, n_classes=oc, n_valid=ov, n_keep=ok)


def conv_case(name, B, H, W, cin, cout, k, down, act, bn, residual, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, cin, generator=g)
    p = {"kernel": torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5}
    if bn:
        p["gamma"] = 1.0 + 0.2 * torch.randn(cout, generator=g)
        p["beta"] = 0.1 * torch.randn(cout, generator=g)
        p["mean"] = 0.1 * torch.randn(cout, generator=g)
        p["var"] = 0.5 + torch.rand(cout, generator=g)
    else:
        p["bias"] = 0.1 * torch.randn(cout, generator=g)
    # operands rounded to bf16 so that the fp32 and bf16 kernels see identical inputs
    x = N.bf16_round(x)
    p["kernel"] = N.bf16_round(p["kernel"])
    y = N.yolo_conv(x, p, down, act)
    out = {"x": x.numpy(), "y_noshort": y.numpy(), "down": np.int32(down), "act": np.array(act or "none"),
           "k": np.int32(k)}
    if residual:
        r = N.bf16_round(torch.randn(*y.shape, generator=g))
        out["residual"] = r.numpy()
        out["y"] = (y + r).numpy()
    else:
        out["y"] = y.numpy()
    for key, v in p.items():
        out["p_" + key] = v.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def main():
    postproc_case("postproc_c3", B=2, sides=(8, 4, 2), C=3, seed=10, conf_bias=1.0, thr=0.25,
                  per_class=4, total=8, iou=0.45)
    postproc_case("postproc_c80", B=2, sides=(13, 7, 4), C=80, seed=11, conf_bias=2.0, thr=0.25,
                  per_class=32, total=32, iou=0.45)
    postproc_case("postproc_c1", B=3, sides=(6, 3, 2), C=1, seed=12, conf_bias=0.5, thr=0.3,
                  per_class=8, total=8, iou=0.3)
    nms_case("nms_dense", B=2, M=600, C=4, seed=20, thr=0.001, per_class=32, total=32, iou=0.45)
    nms_case("nms_ties", B=2, M=300, C=3, seed=21, thr=0.05, per_class=16, total=24, iou=0.3, quant=16)
    conv_case("conv_3x3_mish", 2, 9, 7, 16, 40, 3, False, "mish", False, False, 30)
    conv_case("conv_3x3_down_leaky_bn", 2, 10, 12, 32, 64, 3, True, "leaky", True, False, 31)
    conv_case("conv_1x1_linear", 1, 5, 6, 64, 21, 1, False, None, False, False, 32)
    conv_case("conv_3x3_res", 2, 6, 6, 32, 64, 3, False, "mish", False, True, 33)
    conv_case("conv_first", 1, 12, 12, 3, 32, 3, False, "mish", False, False, 34)
    g = torch.Generator().manual_seed(40)
    x = N.bf16_round(torch.randn(2, 3, 5, 16, generator=g))
    np.savez_compressed(os.path.join(HERE, "upsample.npz"), x=x.numpy(), y=N.upsample2x(x).numpy())


if __name__ == "__main__":
    main()
