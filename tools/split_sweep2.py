"""Stability of the concurrent cut across side-stream instances. Tuning tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd.engine import Detector

B, S = 64, 416
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(B, S, S, 3, device=dev).to(torch.bfloat16)
arg = sys.argv[1] if len(sys.argv) > 1 else "auto"
cut = arg if arg == "auto" else tuple(int(v) for v in arg.split(","))
det = Detector(80, dtype=torch.bfloat16, seed=0, split=cut).to(dev)
det.calibrate_(x[:2].contiguous())


def timeit(steps=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        det(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for inst in range(12):
    det._plans = {}
    det._side_streams = {}
    t0 = time.perf_counter()
    det(x)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    det(x)
    plan = det._plan(B, S, x.dtype, x.device)
    sid = plan.parts[1].stream.stream_id if len(plan.parts) > 1 else -1
    r = [timeit() for _ in range(4)]
    print("instance %2d cut %s stream_id %d plan %.2f s: %s" % (inst, plan.sizes, sid, t_plan, " ".join("%.2f" % v for v in r)), flush=True)
