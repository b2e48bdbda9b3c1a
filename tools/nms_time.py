"""Times decode+filter and the two NMS kernels on the bench's head maps (416x416, batch 64).
YMX_LIB selects another build of the library.  Tuning tool, not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import _lib
if os.environ.get("YMX_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["YMX_LIB"])
from yolo_amd import Detector

dev = torch.device("cuda:0")
B, S = 64, 416
x = torch.rand(B, S, S, 3, generator=torch.Generator().manual_seed(100)).to(dev)
det = Detector(num_classes=80, dtype=torch.bfloat16, seed=0, split="off").to(dev)
det.calibrate_(x[:2].contiguous())
det(x)
plan = det._plan(B, S, x.dtype, dev)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3):
    det.postprocess(plan, S)
e[0].record()
for _ in range(50):
    det.postprocess(plan, S)
e[1].record()
torch.cuda.synchronize()
print("%s postproc %.4f ms  mean survivors %.0f" % (os.environ.get("YMX_LIB", "default"), e[0].elapsed_time(e[1]) / 50,
                                                   float(plan.counts.float().mean())), flush=True)
