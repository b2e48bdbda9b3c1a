"""Runs one convolution layer a few times (target program for rocprofv3 --pmc passes).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/pmc_target.py 64,52,52,512,1024,3,1
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.layers import YOLOConv            # noqa: E402


def main():
    B, H, W, cin, cout, k, stride = (int(v) for v in sys.argv[1].split(","))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, cin, generator=g).cuda().bfloat16()
    layer = YOLOConv((k, k, cout), downsample=(stride == 2), activation="mish", dtype=torch.bfloat16)
    layer.build(cin, generator=g)
    with torch.no_grad():
        layer.kernel.mul_(10.0)
    layer.cuda()
    for _ in range(12):
        layer(x)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
