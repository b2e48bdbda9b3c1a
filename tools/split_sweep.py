"""Times one batch-64 416x416 step for several cuts of the batch into concurrent sub-batches
(and tile thresholds of the shared programs).  Tuning tool, not part of the product."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import engine
from yolo_amd.engine import Detector

B, S = 64, 416
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(B, S, S, 3, device=dev).to(torch.bfloat16)


def run(cut, min_tiles, steps=20, warm=4):
    engine._SHARED_MIN_TILES = min_tiles
    det = Detector(80, dtype=torch.bfloat16, seed=0, split=cut).to(dev)
    det.calibrate_(x[:2].contiguous())
    for _ in range(warm):
        det(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        det(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


cuts = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or \
    [(64,), (48, 16), (46, 18), (50, 14), (47, 17), (49, 15), (48, 8, 8), (48, 12, 4), (40, 24), (44, 12, 8), (48, 16), (64,)]
for cut in cuts:
    for mt in ((80, 120, 50) if len(cut) > 1 else (80,)):
        ms = run(cut, mt)
        print("cut %-14s min_tiles %3d: %.3f ms/step  %.0f img/s" % (cut, mt, ms, B / ms * 1e3), flush=True)
