"""Times the detector's first layer (fp32 image, Cin = 3 -> 32, batch 64, 416x416) with the
three activations, to split its time between the stores and the activation arithmetic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.layers import YOLOConv            # noqa: E402


def main():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(64, 416, 416, 3, generator=g).cuda()
    for act in (None, "leaky", "mish"):
        layer = YOLOConv((3, 3, 32), activation=act, dtype=torch.bfloat16)
        layer.build(3, generator=g)
        with torch.no_grad():
            layer.kernel.mul_(30.0)
        layer.cuda()
        for _ in range(3):
            layer(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            layer(x)
        e1.record()
        torch.cuda.synchronize()
        print("first layer, activation %-5s: %.3f ms" % (act or "none", e0.elapsed_time(e1) / 20), flush=True)


if __name__ == "__main__":
    main()
