"""A/B of one tuning option on the whole detector step, alternating in one process on one box:
    python tools/ab_option.py tile_mix 0 1 [batch] [size]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import Detector, _lib            # noqa: E402

name, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
S = int(sys.argv[5]) if len(sys.argv) > 5 else 416
dev = torch.device("cuda", 0)
det = Detector(num_classes=80, dtype=torch.bfloat16, seed=0, split="off").to(dev)
x = torch.rand(B, S, S, 3, generator=torch.Generator().manual_seed(100)).to(dev)
det.calibrate_(x[:2].contiguous())
L = _lib.lib()


def timed(v, n=60):
    _lib.check(L.ymx_set_option(name, v))
    for _ in range(5):
        det(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        det(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(4):
    a, b = timed(va), timed(vb)
    print("%s=%d: %.4f ms   %s=%d: %.4f ms   (%+.2f %%)" % (name.decode(), va, a, name.decode(), vb, b, (b / a - 1) * 100), flush=True)
