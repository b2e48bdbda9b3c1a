"""Per-launch A/B of one tuning option inside the detector's program, one process, one box:
    python tools/ab_option_perop.py tile_mix 0 1"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from yolo_amd import Detector, _lib            # noqa: E402

name, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
det = Detector(num_classes=80, dtype=torch.bfloat16, seed=0, split="off").to(dev)
x = torch.rand(64, 416, 416, 3, generator=torch.Generator().manual_seed(100)).to(dev)
det.calibrate_(x[:2].contiguous())
plan = det._build_plan((64,), 416, x.dtype, dev)
L = _lib.lib()
res = {}
for rep in range(2):
    for v in (va, vb):
        _lib.check(L.ymx_set_option(name, v))
        bench.profile_ops(plan, x, 2)
        _, per_op = bench.profile_ops(plan, x, 8)
        res.setdefault(v, []).append(per_op)
tot = {va: 0.0, vb: 0.0}
for i in range(len(res[va][0])):
    a = sum(r[i][1] for r in res[va]) / len(res[va])
    b = sum(r[i][1] for r in res[vb]) / len(res[vb])
    tot[va] += a
    tot[vb] += b
    na, nb = res[va][0][i][0], res[vb][0][i][0]
    if na != nb or abs(b / a - 1) > 0.02:
        print("%3d %-36s %-36s %s  %.4f -> %.4f ms (%+.1f %%)" % (i, na, nb, res[va][0][i][3], a, b, (b / a - 1) * 100))
print("sum of launches: %.4f -> %.4f ms" % (tot[va], tot[vb]))
