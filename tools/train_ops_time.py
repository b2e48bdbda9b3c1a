"""Times the kernels beyond plain inference at training-sized inputs and prices them
against their algorithmic HBM bytes (run on the GPU box):
    python tools/train_ops_time.py [batch]
yolo loss forward per scale (reads conv + label rows once), SPP block (reads x, writes 4x),
label assignment (writes the label tensors), CSPDarknet53 forward (bf16)."""
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_amd as Y                                                     # noqa: E402
from yolo_amd import losses as L, dataset as DS                          # noqa: E402
from yolo_amd.layers import spp_concat                                   # noqa: E402
from yolo_amd.backbones import CSPDarknet53                              # noqa: E402

DEV = "cuda:0"
PEAK = 8000.0


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    C, size = 80, 416
    cfg = Y.config
    rng = np.random.default_rng(0)
    nmax = 64
    counts = rng.integers(1, nmax, B).astype(np.int32)
    boxes = np.zeros((B, nmax, 5), np.float32)
    for b in range(B):
        n = counts[b]
        wh = np.exp(rng.normal(3.5, 1.0, (n, 2))).clip(4, size * 0.9)
        ctr = rng.uniform(0, size, (n, 2))
        x0 = np.clip(ctr - wh / 2, 0, size - 2)
        x1 = np.clip(ctr + wh / 2, x0 + 1, size - 1)
        boxes[b, :n] = np.floor(np.concatenate([x0, x1, rng.integers(0, C, (n, 1))], 1))
    bx, ct = torch.from_numpy(boxes).to(DEV), torch.from_numpy(counts).to(DEV)
    ms = timed(lambda: DS.preprocess_true_boxes(bx, ct, size, C))
    labels = DS.preprocess_true_boxes(bx, ct, size, C)
    lab_bytes = sum(t.numel() * 4 for t in labels)
    print("assign_labels  B=%d: %.3f ms (incl. clearing %.1f MB of labels: %.0f GB/s written)" %
          (B, ms, lab_bytes / 1e6, lab_bytes / ms / 1e6))
    tg = DS.pack_targets(*labels)
    for scale in (0, 1, 2):
        S = size // cfg["strides"][scale]
        out = (torch.randn(B, S, S, 3 * (5 + C), generator=torch.Generator().manual_seed(scale)) * 2).to(DEV)
        f = lambda: L.loss_at_scale(tg[scale], out, scale, C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"], terms=True)
        ms = timed(f)
        alg = 2 * B * S * S * 3 * (5 + C) * 4 + B * 128 * 4 * 4
        print("yolo_loss  S=%d B=%d: %.3f ms, %.1f MB algorithmic -> %.0f GB/s = %.3f of 8 TB/s" %
              (S, B, ms, alg / 1e6, alg / ms / 1e6, alg / ms / 1e6 / PEAK))
        og = out.clone().requires_grad_(True)
        w3 = torch.ones(3, device=DEV)

        def fb():
            og.grad = None
            (L.loss_at_scale(tg[scale], og, scale, C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"], terms=True) * w3).sum().backward()
        ms2 = timed(fb)
        algb = alg + B * S * S * 3 * (5 + C) * 4
        print("yolo_loss forward + gradient S=%d: %.3f ms (gradient part %.3f ms, %.1f MB algorithmic incl. the gradient written -> %.3f of 8 TB/s)" %
              (S, ms2, ms2 - ms, algb / 1e6, algb / max(ms2 - ms, 1e-6) / 1e6 / PEAK))
    for H, Cc, dt in ((13, 512, torch.bfloat16), (19, 512, torch.bfloat16), (13, 512, torch.float32)):
        x = torch.randn(B, H, H, Cc, device=DEV).to(dt)
        ms = timed(lambda: spp_concat(x))
        alg = x.numel() * x.element_size() * 5
        print("spp_concat  %dx%dx%d %s B=%d: %.3f ms, %.1f MB -> %.0f GB/s = %.3f of 8 TB/s" %
              (H, H, Cc, str(dt).split(".")[1], B, ms, alg / 1e6, alg / ms / 1e6, alg / ms / 1e6 / PEAK))
    net = CSPDarknet53(dtype=torch.bfloat16)
    net.to(DEV)
    img = torch.rand(B, size, size, 3, device=DEV)
    t0 = time.time()
    net(img)
    torch.cuda.synchronize()
    first = time.time() - t0
    ms = timed(lambda: net(img), reps=5, warm=1)
    print("CSPDarknet53 bf16 module walk B=%d %d^2: %.2f ms per forward (%.0f img/s; first call %.2f s)" %
          (B, size, ms, B / ms * 1e3, first))


if __name__ == "__main__":
    main()
