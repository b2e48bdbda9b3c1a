"""Diagnostic for the decode kernel's box corners under concurrent streams.

Runs the detector on a batch cut into two concurrent sub-batches (the schedule under which a
build of the decode kernel with packed-FMA box corners, `make -C yolo_amd/csrc packed-decode`, produces wrong corners) and, for every
64-candidate segment whose boxes differ from the whole-batch program's, prints what the wrong
values are made of: the segment, the lanes and components affected, and which closed form
(cy/size, (cy -+ bw/2)/size, ...) each wrong value equals, recomputed here from the head maps.
    YMX_LIBRARY=yolo_amd/libymx_packed_decode.so REPS=40 python tools/corner_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import Detector
from yolo_amd.config import config

DEV = "cuda:0"
C = 80
REPS = int(os.environ.get("REPS", "40"))
SIZE = int(os.environ.get("SIZE", "160"))
B = 7
Y = config["yolo"]
x = torch.rand(B, SIZE, SIZE, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
whole = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split="off").to(DEV)
whole.calibrate_(x[:2].contiguous())
whole(x)
torch.cuda.synchronize()
wplan = whole._plan(B, SIZE, x.dtype, x.device)
wp = wplan.parts[0]
wboxes, wcounts, wsrc = wp.boxes.clone(), wp.counts.clone(), wp.src_idx.clone()
heads = [h.clone().float() for h in wplan.heads]              # [B,S,S,255] fp32

# closed forms per candidate, from the head maps (float32, same formulas as utils.py:107-167)
anc = torch.tensor(Y["anchors"], dtype=torch.float32).reshape(3, 3, 2).to(DEV)
forms = []
for i, h in enumerate(heads):
    S = h.shape[1]
    t = h.reshape(B, S, S, 3, 5 + C)
    xs, st = float(Y["xyscale"][i]), float(Y["strides"][i])
    col = torch.arange(S, device=DEV, dtype=torch.float32).view(1, 1, S, 1)
    row = torch.arange(S, device=DEV, dtype=torch.float32).view(1, S, 1, 1)
    cx = (torch.sigmoid(t[..., 0]) * xs - 0.5 * (xs - 1) + col) * st
    cy = (torch.sigmoid(t[..., 1]) * xs - 0.5 * (xs - 1) + row) * st
    bw = torch.exp(t[..., 2]) * anc[i, :, 0].view(1, 1, 1, 3)
    bh = torch.exp(t[..., 3]) * anc[i, :, 1].view(1, 1, 1, 3)
    forms.append(torch.stack([cx, cy, bw, bh], -1).reshape(B, -1, 4))
forms = torch.cat(forms, 1)                                   # [B,N,4] (cx,cy,bw,bh)


def explain(val, f):
    cx, cy, bw, bh = [float(v) for v in f]
    cands = {"ymin": (cy - bh / 2) / SIZE, "xmin": (cx - bw / 2) / SIZE, "ymax": (cy + bh / 2) / SIZE,
             "xmax": (cx + bw / 2) / SIZE, "cy/S": cy / SIZE, "cx/S": cx / SIZE, "(cy-bw/2)/S": (cy - bw / 2) / SIZE,
             "(cy+bw/2)/S": (cy + bw / 2) / SIZE, "bh/2/S": bh / 2 / SIZE, "(cy-bh)/S": (cy - bh) / SIZE,
             "(cy+bh)/S": (cy + bh) / SIZE, "(cy-bh/2)": cy - bh / 2, "(cy+bh/2)": cy + bh / 2, "0": 0.0}
    hits = [k for k, v in cands.items() if abs(val - v) <= 2e-6 * max(1.0, abs(v))]
    return hits or ["?"]


bad_total = 0
for cut in ((4, 3), (5, 2)):
    det = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split=cut).to(DEV)
    det.calibrate_(x[:2].contiguous())
    for rep in range(REPS):
        det(x)
        torch.cuda.synchronize()
        plan = det._plan(B, SIZE, x.dtype, x.device)
        start = 0
        for pi, part in enumerate(plan.parts):
            for bl in range(part.B):
                bg = start + bl
                n = int(wcounts[bg])
                if int(part.counts[bl]) != n:
                    print("cut", cut, "rep", rep, "image", bg, "COUNT", int(part.counts[bl]), n, flush=True)
                    continue
                got, want = part.boxes[bl, :n], wboxes[bg, :n]
                src = wsrc[bg, :n].long()
                if not torch.equal(part.src_idx[bl, :n].long(), src):
                    print("cut", cut, "rep", rep, "image", bg, "SRC_IDX differs", flush=True)
                diff = got != want
                if not bool(diff.any()):
                    continue
                rows = diff.any(1).nonzero().reshape(-1)
                segs = torch.unique(src[rows] // 64).tolist()
                for sg in segs:
                    in_seg = (src // 64) == sg
                    pos = in_seg.nonzero().reshape(-1)
                    d = diff[pos]
                    bad_total += 1
                    comp = d.any(0).tolist()
                    print("cut", cut, "rep", rep, "part", pi, "image", bg, "segment", sg, "of", (forms.shape[1] + 63) // 64,
                          "survivors", len(pos), "bad lanes", int(d.any(1).sum()), "components[ymin,xmin,ymax,xmax]", comp, flush=True)
                    print("  bad lanes:", sorted(int(src[p]) % 64 for p in pos.tolist() if bool(diff[p].any())),
                          "| good survivor lanes:", sorted(int(src[p]) % 64 for p in pos.tolist() if not bool(diff[p].any())), flush=True)
                    shown = 0
                    for p in pos.tolist():
                        if not bool(diff[p].any()) or shown >= 3:
                            continue
                        shown += 1
                        f = forms[bg, int(src[p])]
                        line = ["  lane", int(src[p]) % 64]
                        for k in range(4):
                            if bool(diff[p, k]):
                                line += ["c%d got %.7g want %.7g =" % (k, float(got[p, k]), float(want[p, k]))] + explain(float(got[p, k]), f)
                        print(*line, flush=True)
            start += part.B
print("bad segments", bad_total, "in", 2 * REPS, "concurrent forwards; done")
