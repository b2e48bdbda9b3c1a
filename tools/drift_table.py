"""Where does the bf16 network drift from the fp32 one?  (run on the GPU)

Both networks are the detector's Darknet53 + YoloV3 with the same activation-calibrated random
weights, run layer by layer through the module path on the same images; the fp32 network runs
the fp32 MFMA kernels of libymx.  Printed per convolution (execution order): the relative RMS
difference of its output, bf16 network against fp32 network.  Then the experiment: the
residual stream of every ResidialBlock kept in fp32 (the 3x3 convolution writes fp32, the
shortcut is added in fp32, only the convolutions' inputs are rounded to bf16) and the same
table for the three head outputs.
    python tools/drift_table.py [batch] [size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import Detector                                           # noqa: E402
from yolo_amd.layers import YOLOConv, ResidialBlock                     # noqa: E402

DEV = "cuda:0"


def rel_rms(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).square().mean() / b.square().mean().clamp_min(1e-300)).sqrt())


def run(det, x, keep):
    hooks = []
    for name, m in list(det.backbone.named_modules()) + [("head." + n, m) for n, m in det.head.named_modules()]:
        if isinstance(m, YOLOConv):
            hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: keep.append((name, mod, out.float().cpu()))))
    with torch.no_grad():
        dt = det.head.conv_layers()[0].compute_dtype
        heads = det.head(*[s.to(dt) for s in det.backbone(x)])       # (the stage outputs enter the head in its own type)
    for h in hooks:
        h.remove()
    return [h.float().cpu() for h in heads]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
    x = torch.rand(B, S, S, 3, generator=torch.Generator().manual_seed(100)).to(DEV)
    det_b = Detector(num_classes=80, dtype=torch.bfloat16, seed=0).to(DEV)
    det_b.calibrate_(x[:2].contiguous())
    det_f = Detector(num_classes=80, dtype=torch.float32, seed=0).to(DEV)
    det_f.load_state_dict(det_b.state_dict())
    kb, kf = [], []
    hb = run(det_b, x, kb)
    hf = run(det_f, x, kf)
    assert len(kb) == len(kf), (len(kb), len(kf))
    print("%-3s %-34s %-5s %-12s %s" % ("#", "layer", "k/s", "channels", "rel. RMS difference bf16 vs fp32"))
    rows = []
    for i, ((name, mod, ob), (_, _, of)) in enumerate(zip(kb, kf)):
        e = rel_rms(ob, of)
        rows.append((i, name, e))
        print("%-3d %-34s %d/%d   %4d->%-5d  %.3e" % (i, name, mod.kernel_size, 2 if mod.downsample else 1, mod.in_channels, mod.filters, e))
    base = [rel_rms(a, b) for a, b in zip(hb, hf)]
    print("head outputs (52x52, 26x26, 13x13):", " ".join("%.3e" % e for e in base))

    # the experiment: fp32 residual stream
    orig = ResidialBlock.forward

    def fp32_stream(self, x, out=None):
        h = self.layer_stack[0](x)                                   # rounds its input to bf16
        y = self.layer_stack[1](h, out_dtype=torch.float32)
        return y + x.float()

    ResidialBlock.forward = fp32_stream
    try:
        ke = []
        he = run(det_b, x, ke)
    finally:
        ResidialBlock.forward = orig
    exp = [rel_rms(a, b) for a, b in zip(he, hf)]
    print("head outputs with the residual stream in fp32:", " ".join("%.3e" % e for e in exp))
    print("ratio (fp32 stream / bf16 stream):", " ".join("%.2f" % (a / b) for a, b in zip(exp, base)))


if __name__ == "__main__":
    main()
