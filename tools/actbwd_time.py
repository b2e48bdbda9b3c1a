"""Times ymx_act_backward_bias on the detector's activation shapes at batch 64 (tuning aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib
L = _lib.lib()
DEV = "cuda:0"
tot = 0.0
for hw, c, n in ((416, 32, 1), (208, 64, 2), (208, 32, 1), (104, 128, 3), (104, 64, 2), (52, 256, 9), (52, 128, 8), (26, 512, 9), (26, 256, 8), (52, 1024, 3), (52, 512, 4), (13, 1024, 5)):
    P = 64 * hw * hw
    z = torch.randn(P, c, device=DEV).bfloat16(); dy = torch.randn(P, c, device=DEV).bfloat16(); dz = torch.empty_like(z)
    db = torch.empty(c, device=DEV)
    nb = L.ymx_act_backward_bias_workspace_bytes(P, c, 2, _lib.BF16)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    f = lambda: _lib.launch("ymx_act_backward_bias", z, _lib.ptr(z), _lib.ptr(dy), _lib.ptr(dz), P, c, 2, _lib.BF16, _lib.ptr(db), _lib.ptr(ws), nb)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    tot += ms * n
    print("%3d^2 x %4d: %.4f ms  %.2f TB/s" % (hw, c, ms, 3 * P * c * 2 / ms / 1e9))
print("weighted total %.3f ms" % tot)
