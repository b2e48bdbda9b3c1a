"""Does running the batch as S sub-batches on S HIP streams fill the tail of each launch?

Times one batch-64 step (forward + post-processing) as 1 x 64, 2 x 32 and 4 x 16 images,
each sub-batch on its own stream and its own Detector (same seed, so same weights).
Tuning tool, not part of the product.
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd.engine import Detector

B, S = 64, 416
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(B, S, S, 3, device=dev).to(torch.bfloat16)
offset = int(os.environ.get("OFFSET", "0"))


from yolo_amd import _lib


def run(sizes, min_tiles=160, prio=None, steps=20, warm=5):
    _lib.check(_lib.lib().ymx_set_option(b"dma_min_tiles", min_tiles))
    nsplit = len(sizes)
    dets = [Detector(80, dtype=torch.bfloat16, seed=0).to(dev) for _ in range(nsplit)]
    for d in dets:
        d.calibrate_(x[:4])
    streams = [torch.cuda.Stream(priority=(prio[i] if prio else 0)) for i in range(nsplit)]
    starts = [sum(sizes[:i]) for i in range(nsplit)]
    xs = [x[a:a + n].contiguous() for a, n in zip(starts, sizes)]
    main = torch.cuda.current_stream()

    def step():
        ev = torch.cuda.Event()
        ev.record(main)
        for d, st, xi in zip(dets, streams, xs):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                d(xi)
        for st in streams:
            e = torch.cuda.Event()
            e.record(st)
            main.wait_event(e)

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "", flush=True)
for sizes, prio in (((64,), None), ((48, 16), None), ((48, 16), (-1, 0)), ((48, 16), (0, -1)), ((48, 8, 8), None), ((48, 8, 8), (-1, 0, 0)),
                    ((32, 16, 16), None), ((48, 16), None), ((64,), None)):
    for mt in (80,):
        ms = run(sizes, mt, prio)
        print("split %-15s prio %-12s min_tiles %3d: %.3f ms/step  %.0f img/s" % (sizes, prio, mt, ms, B / ms * 1e3), flush=True)
