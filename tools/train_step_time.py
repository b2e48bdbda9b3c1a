"""Times one training step of the detector on one GPU (tuning aid):
forward with the graph recorded, the YOLO loss at the three scales, backward through every
block, plain SGD.      python tools/train_step_time.py [batch] [size] [steps]
env: LR, WGRAD (option wgrad_variant), OVERLAP (0: weight gradients on the main stream)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_amd as Y                                                     # noqa: E402
from yolo_amd import losses as L, dataset as DS, training as T          # noqa: E402
from yolo_amd.backbones import Darknet53                                 # noqa: E402
from yolo_amd.detectors import YoloV3                                    # noqa: E402

DEV = "cuda:0"


def make_targets(B, size, C, rng):
    nmax = 32
    counts = rng.integers(1, nmax, B).astype(np.int32)
    boxes = np.zeros((B, nmax, 5), np.float32)
    for b in range(B):
        n = counts[b]
        wh = np.exp(rng.normal(3.5, 1.0, (n, 2))).clip(4, size * 0.9)
        ctr = rng.uniform(0, size, (n, 2))
        x0 = np.clip(ctr - wh / 2, 0, size - 2)
        x1 = np.clip(ctr + wh / 2, x0 + 1, size - 1)
        boxes[b, :n] = np.floor(np.concatenate([x0, x1, rng.integers(0, C, (n, 1))], 1))
    labels = DS.preprocess_true_boxes(torch.from_numpy(boxes).to(DEV), torch.from_numpy(counts).to(DEV), size, C)
    return DS.pack_targets(*labels)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    C = 80
    cfg = Y.config
    if os.environ.get("WGRAD"):
        T.options["wgrad_variant"] = int(os.environ["WGRAD"])      # applied on autograd's thread, where the launches happen
    det = Y.Detector(num_classes=C, dtype=torch.bfloat16, seed=0).to(DEV)
    x = torch.rand(B, size, size, 3, device=DEV)
    det.calibrate_(x[:2].contiguous())
    backbone, head = det.backbone, det.head
    params = [p for p in list(backbone.parameters()) + list(head.parameters())]
    fns = L.get_yolo_loss_at_scales([0, 1, 2], C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"])   # order: scales 2, 1, 0
    targets = make_targets(B, size, C, np.random.default_rng(0))
    lr = float(os.environ.get("LR", "1e-7"))
    if os.environ.get("OVERLAP"):
        T.overlap_wgrad = bool(int(os.environ["OVERLAP"]))

    reg = T.L2Regulariser(det)

    def step():
        with T.backprop():
            s, m, l = head(*backbone(x))
        loss = fns[0](targets[2], l) + fns[1](targets[1], m) + fns[2](targets[0], s)
        for p in params:
            p.grad = None
        loss.backward()
        reg.add_gradients_()
        with torch.no_grad():
            torch._foreach_add_(params, [p.grad for p in params], alpha=-lr)
        return loss

    losses = [float(step().detach())]
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    losses.append(float(loss.detach()))
    print("train step B=%d %d^2: %.1f ms/step = %.0f img/s; loss %.3f -> %.3f; peak memory %.1f GB" %
          (B, size, dt * 1e3, B / dt, losses[0], losses[-1], torch.cuda.max_memory_allocated() / 1e9))


if __name__ == "__main__":
    main()
