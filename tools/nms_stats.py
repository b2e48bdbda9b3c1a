"""Candidate statistics of the bench workload's NMS stage (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.engine import Detector            # noqa: E402


def main():
    torch.manual_seed(0)
    det = Detector(num_classes=80, dtype=torch.bfloat16, seed=0, split="off").cuda()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(64, 416, 416, 3, generator=g).cuda()
    det.calibrate_(x[:8])
    out = det.forward(x)
    torch.cuda.synchronize()
    plan = det._plan(64, 416, x.dtype, x.device)
    counts = plan.counts.cpu()
    sc = plan.parts[0].scores_cm              # [B, C, cap]
    thr = det.score_threshold
    print("score_thr", thr, "iou_thr", det.iou_threshold, "per_class", det.max_outputs_per_class, "total", det.max_outputs)
    print("survivors/image: mean %.0f max %d cap %d" % (counts.float().mean(), counts.max(), plan.cap))
    per = []
    for b in range(0, 64, 8):
        m = int(counts[b])
        n = (sc[b, :, :m] > thr).sum(dim=1)
        per.append(n)
    per = torch.cat(per).float().cpu()
    print("candidates per (image, class) above thr: mean %.1f median %.0f p90 %.0f max %.0f; fraction zero %.2f" % (
        per.mean(), per.median(), per.quantile(0.9), per.max(), (per == 0).float().mean()))
    print("valid detections/image: mean %.1f" % out.valid.float().mean())


if __name__ == "__main__":
    main()
