"""Repeats the detector on a batch cut into two concurrent sub-batches and compares, per image,
the decode+filter output (boxes, source indices, counts) and the final detections with the
whole-batch program's; then the second sub-batch's images alone.  Prints every difference.
    REPS=100 python tools/concurrency_screen.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import Detector
DEV = "cuda:0"
C = 80
REPS = int(os.environ.get("REPS", "80"))
x = torch.rand(7, 160, 160, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
whole = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split="off").to(DEV)
whole.calibrate_(x[:2].contiguous())
want = whole(x)
torch.cuda.synchronize()
wp = whole._plan(7, 160, x.dtype, x.device).parts[0]
wboxes, wcounts, wsrc = wp.boxes.clone(), wp.counts.clone(), wp.src_idx.clone()
wout = want[0].clone()
for cut in ((4, 3), (5, 2)):
    det = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split=cut).to(DEV)
    det.calibrate_(x[:2].contiguous())
    for rep in range(REPS):
        got = det(x)
        torch.cuda.synchronize()
        plan = det._plan(7, 160, x.dtype, x.device)
        start = 0
        for pi, part in enumerate(plan.parts):
            for bl in range(part.B):
                bg = start + bl
                n = int(wcounts[bg])
                if int(part.counts[bl]) != n:
                    print("cut", cut, "rep", rep, "image", bg, "count", int(part.counts[bl]), n)
                    continue
                db = (part.boxes[bl, :n] != wboxes[bg, :n]).any(dim=1)
                ds = (part.src_idx[bl, :n] != wsrc[bg, :n])
                if bool(db.any()) or bool(ds.any()):
                    idx = db.nonzero().reshape(-1)[:6].tolist()
                    print("cut", cut, "rep", rep, "part", pi, "image", bg, "decode boxes differ at positions", idx, "n", n,
                          "src_idx diff", int(ds.sum()), "got", part.boxes[bl, idx[0]].tolist(), "want", wboxes[bg, idx[0]].tolist())
            start += part.B
        if not torch.equal(got[0], wout):
            print("cut", cut, "rep", rep, "final boxes differ")
# the second part's images alone, whole-batch program, no concurrency
alone = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split="off").to(DEV)
alone.calibrate_(x[:2].contiguous())
xs = x[4:7].contiguous()
for rep in range(REPS):
    g = alone(xs)
    torch.cuda.synchronize()
    if not torch.equal(g[0], wout[4:7]):
        print("alone rep", rep, "final boxes differ")
print("done")
