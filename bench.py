"""Benchmark of the detector hot path: images/sec for forward + decode + filter + NMS.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload: Darknet53 + YoloV3 (80 classes), 416x416, batch 64 per GPU, bf16 storage with
fp32 accumulation, random weights re-scaled so activations stay O(1) (random data, not
zeros), synthetic images already resident in HBM.  One step = one batch through all 73
convolutions, the two upsample/concats, decode+filter and combined NMS; with N > 1 every
rank runs its own 64 images (weak scaling) and the step ends with the all-gather of the
detection records.  The engine may run a rank's batch as two concurrent sub-batches on
separate HIP streams (config.sub_batches; same results bit for bit).  Rank 0 prints ONE
JSON line.

Besides the headline value the line carries
  roofline      the kernel with the most time per step (the resident-slab 3x3 bf16
                implicit-GEMM convolution): algorithmic FLOPs of its launches / their summed
                duration, measured with events between the launches of the whole batch run
                as one program (each launch alone on the GPU), against the
                dense bf16 MFMA peak; `traffic` = HBM bytes per launch of that kernel from
                the committed rocprofv3 --pmc passes (profiles/), null if there are none;
  cpu_baseline  the CPU oracle (torch-CPU fp32 convs + C post-processing) timed on this
                host on a bounded sample of the same workload (rank 0, N = 1 only);
  breakdown.stress / breakdown.train   (single GPU, default shape) short legs of the NMS-stress
                and training-step workloads -- `--workload nms-stress` / `--workload train` run
                alone -- each with value, ms_per_step, roofline and its own cpu_baseline.
The K timed steps are repeated (`repeats`) so that the timed region lasts about two seconds;
value and ms_per_step are per step over all of them.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak of one MI355X (spec)
PEAK_HBM_GBS = 8000.0          # HBM3E peak (spec)


def conv_kernel_name(op):
    """Label of the kernel the library launches for a program op (asked of the library itself)."""
    if op.kind != 1:
        return "upsample2x_concat"
    from yolo_amd import _lib
    fam = _lib.lib().ymx_conv_kernel_label(_lib.ctypes.byref(op))
    if fam is None:
        raise RuntimeError("bench: " + _lib.lib().ymx_last_error().decode())
    fam = fam.decode()
    bf16 = op.dtype == 1
    if fam in ("halo", "dma", "dma4"):
        return "conv_igemm_%s_bf16_256x256" % fam
    if fam == "halo192":
        return "conv_igemm_halo_bf16_256x192"
    if fam == "halo_mixed":
        return "conv_igemm_halo_bf16_256x256+192"
    if fam == "dma4_mixed":
        return "conv_igemm_dma4_bf16_256x256+192"
    if fam == "dma4_192":
        return "conv_igemm_dma4_bf16_256x192"
    if fam == "slab2":
        return "conv_igemm_slab2_bf16_cin%d" % op.Cin
    if fam == "dma2":
        tq = 32 if op.Cout <= 32 else (64 if op.Cout <= 64 else (128 if op.Cout <= 128 else 256))
        return "conv_igemm_dma2_bf16_%dx256" % tq
    if fam == "igemm":
        tile = "256x32" if op.Cout <= 32 else ("256x64" if op.Cout <= 64 else "128x128")
        return "conv_igemm_%s_%s" % ("bf16" if bf16 else "f32", tile)
    return "conv_" + fam


# bench label -> device kernel name in the rocprofv3 summaries under profiles/
PMC_KERNEL = {
    "conv_igemm_halo_bf16_256x256": "conv_igemm_halo_kernel<bf16, 4>",
    "conv_igemm_halo_bf16_256x192": "conv_igemm_halo_kernel<bf16, 3>",
    "conv_igemm_halo_bf16_256x256+192": "conv_igemm_halo_mixed_kernel<bf16>",
    "conv_igemm_dma4_bf16_256x256+192": "conv_igemm_dma4_mixed_kernel<bf16>",
    "conv_igemm_dma_bf16_256x256": "conv_igemm_dma_kernel<bf16, 256>",
    "conv_igemm_dma4_bf16_256x256": "conv_igemm_dma4_kernel<bf16, 4>",
    "conv_igemm_dma4_bf16_256x192": "conv_igemm_dma4_kernel<bf16, 3>",
    "conv_stem_fused": "conv_stem_kernel<2>",
    "conv_resblock_fused": "conv_resblock_kernel<2>",
}


def pmc_traffic(label, pattern="r*_pmc_traffic.json", kernel=None):
    """HBM bytes per launch of a kernel, from the newest committed --pmc summary matching
    ``pattern`` under profiles/ (None when there is none)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    kernel = kernel or PMC_KERNEL.get(label)
    if not files or kernel is None:
        return None
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"].get(kernel)
        return None if k is None else round(k["hbm_bytes_per_launch"], 1)
    except (OSError, ValueError, KeyError):
        return None


def _sum_traffic(pattern, prefixes):
    """Summed HBM bytes per step of every kernel whose name starts with one of ``prefixes``."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            ks = json.load(f)["kernels"]
        tot = [v["hbm_bytes_per_launch"] * v.get("launches_per_step", 1.0) for k, v in ks.items() if k.startswith(prefixes)]
        return round(sum(tot), 1) if tot else None
    except (OSError, ValueError, KeyError):
        return None


def time_postproc_split(run_decode, run_nms, steps):
    """decode+filter and NMS timed apart with events between the two calls on the stream they
    run on (ms per call each)."""
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    run_decode(); run_nms()
    torch.cuda.synchronize()
    for e in ev:
        e[0].record(); run_decode(); e[1].record(); run_nms(); e[2].record()
    torch.cuda.synchronize()
    dec = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
    nms = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
    return dec, nms


def hbm_roofline(kernel, bytes_per_launch, ms, traffic):
    gbs = bytes_per_launch / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": round(bytes_per_launch, 1),
            "avg_launch_ms": round(ms, 4)}


def postproc_bytes(B, N, C, m_mean, head_bytes, total=32):
    """Algorithmic bytes of one decode+filter call and of one combined-NMS call over B images:
    decode reads every logit once and writes box + class-major score column + source index
    per survivor (+ the count); NMS reads each survivor's box and scores once and writes the
    detection records."""
    dec = B * (N * (5 + C) * head_bytes + m_mean * (4 + C + 1) * 4 + 4)
    nms = B * (m_mean * (4 + C) * 4 + total * 6 * 4 + 4 + total * 4)
    return dec, nms


def stress_workload(args, dev):
    """BASELINE config 4: decode + filter + combined NMS alone on synthetic head maps of a
    1280x1280 input (100 800 candidates per image, 80 classes, score threshold 0.001).
    Returns the result line (a dict)."""
    from yolo_amd import _lib
    from yolo_amd import utils as U
    import yolo_amd as Y
    B, C = args.batch, args.classes
    sides = (160, 80, 40)
    heads = []
    for seed, S in zip((10, 11, 12), sides):
        g = torch.Generator(device="cpu").manual_seed(seed)
        h = torch.empty(B, S, S, 3, 5 + C)
        for i in range(B):                           # image by image: bounded host memory
            h[i] = torch.randn(S, S, 3, 5 + C, generator=g) * 2.0
        h[..., 4] += 6.0                             # objectness bias: nearly every candidate passes 0.001
        heads.append(h.to(dev))
    cfg = Y.config
    sc = U._Scales(heads, cfg["strides"], cfg["xyscale"], U.default_anchors())
    N = sc.N
    cap = U._round4(N)
    f32, i32 = torch.float32, torch.int32
    boxes = torch.empty((B, cap, 4), device=dev, dtype=f32)
    scores = torch.empty((B, C, cap), device=dev, dtype=f32)
    src = torch.empty((B, cap), device=dev, dtype=i32)
    counts = torch.empty((B,), device=dev, dtype=i32)
    T = 32
    outs = [torch.empty((B, T, 4), device=dev, dtype=f32), torch.empty((B, T), device=dev, dtype=f32),
            torch.empty((B, T), device=dev, dtype=f32), torch.empty((B,), device=dev, dtype=i32),
            torch.empty((B, T), device=dev, dtype=i32)]
    L = _lib.lib()
    dws_n = L.ymx_decode_workspace_bytes(B, sc.n, sc.S)
    dws = torch.empty((dws_n,), device=dev, dtype=torch.uint8)
    nws_n = L.ymx_nms_workspace_bytes(B, C, 32)
    nws = torch.empty((nws_n,), device=dev, dtype=torch.uint8)
    thr, size = 0.001, 1280.0

    def run_decode():
        _lib.launch("ymx_decode_filter", boxes, sc.ptrs, sc.n, sc.S, B, C, _lib.F32, sc.anchors, sc.strides, sc.xyscale,
                    1, thr, size, _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(src), _lib.ptr(counts), cap,
                    _lib.ptr(dws), dws_n)

    def run_nms():
        _lib.launch("ymx_nms_combined", boxes, _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(counts), _lib.ptr(src), B, cap,
                    C, 32, T, 0.45, thr, *[_lib.ptr(t) for t in outs], _lib.ptr(nws), nws_n)

    def step():
        run_decode(); run_nms()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / args.steps * 1e3
    m_mean = float(counts.float().mean().item())
    dec_ms, nms_ms = time_postproc_split(run_decode, run_nms, max(3, min(args.steps, 20)))
    dec_b, nms_b = postproc_bytes(B, N, C, m_mean, 4)
    pat = "r*_pmc_traffic_stress.json"
    line = {
        "metric": "images/sec (decode+filter+NMS only) NMS stress", "value": round(B * args.steps / elapsed, 2),
        "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "candidates_per_s": round(B * N * args.steps / elapsed, 1),
        "config": {"workload": "NMS stress: decode + filter + combined NMS on synthetic fp32 head maps "
                               "[%d,160,160,3,%d] [..,80,80,..] [..,40,40,..] (N(0,2^2) logits, objectness +6, seeds 10/11/12), "
                               "%d candidates per image, %d classes, score threshold 0.001, IoU 0.45, 32 per class, 32 total"
                               % (B, 5 + C, N, C),
                   "batch": B, "candidates_per_image": N, "num_classes": C, "mean_survivors_per_image": m_mean,
                   "mean_valid_detections": float(outs[3].float().mean().item())},
        "roofline": hbm_roofline("decode_kernel<float, 20, 1, true>", dec_b, dec_ms, pmc_traffic(None, pat, "decode_kernel<float, 20, 1, true>")),
        "roofline_nms": hbm_roofline("nms_class_kernel + nms_merge_kernel", nms_b, nms_ms, _sum_traffic(pat, ("nms_",))),
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_stress(C)
    return line


def cpu_baseline_stress(C, budget_s=15.0):
    """The C oracle's decode + filter + NMS on one stress image at a time, on this host."""
    import numpy as np
    from oracle import cref
    rng = np.random.default_rng(10)
    heads = []
    for S in (160, 80, 40):
        h = (rng.standard_normal((1, S, S, 3, 5 + C)) * 2.0).astype(np.float32)
        h[..., 4] += 6.0
        heads.append(h)
    cref.bbox_nms(heads, 1280.0, C, 32, 32, 0.45, 0.001)
    t0 = time.perf_counter()
    n = 0
    while True:
        cref.bbox_nms(heads, 1280.0, C, 32, 32, 0.45, 0.001)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 64:
            break
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": "%d stress images (100 800 candidates, 80 classes) through the C oracle's decode/filter/NMS, one thread (%.1f s)" % (n, dt)}


def conv_flops(op):
    if op.kind != 1:
        return 0.0
    oh = op.H if op.stride == 1 else (op.H + 1 - op.k) // 2 + 1
    ow = op.W if op.stride == 1 else (op.W + 1 - op.k) // 2 + 1
    return 2.0 * op.B * oh * ow * op.k * op.k * op.Cin * op.Cout


def profile_ops(plan, x, steps):
    """Per-launch durations of a whole-batch program, each launch alone on the GPU: events
    between the launches on the stream they run on."""
    from yolo_amd import _lib
    plan = plan.parts[0]
    plan.ops[0].x = x.data_ptr()
    L = _lib.lib()
    # launch units: one op each, except where the program marks a convolution as read by the next op alone
    # (ymx_op.C1): ymx_run_program runs such a pair as one fused launch (the first two layers, the first ResidialBlock)
    units, i = [], 0
    while i < plan.n_ops:
        cnt = 2 if (plan.ops[i].kind == 1 and plan.ops[i].C1 != 0 and i + 1 < plan.n_ops) else 1
        units.append((i, cnt))
        i += cnt
    n = len(units)
    acc = [0.0] * n
    elems = [plan.ops[i] for i in range(plan.n_ops)]             # views into the program array
    op_ptr = [_lib.ctypes.byref(elems[i]) for i, _ in units]
    for _ in range(steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        for u, (i, cnt) in enumerate(units):
            ev[u].record()
            _lib.check(L.ymx_run_program(op_ptr[u], cnt, _lib.stream()))
        ev[n].record()
        torch.cuda.synchronize()
        for u in range(n):
            acc[u] += ev[u].elapsed_time(ev[u + 1])
    groups = {}
    per_op = []
    for u, (i, cnt) in enumerate(units):
        name = ("conv_stem_fused" if plan.ops[i].Cin == 3 else "conv_resblock_fused") if cnt == 2 else conv_kernel_name(plan.ops[i])
        flops = sum(conv_flops(plan.ops[j]) for j in range(i, i + cnt))
        g = groups.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
        g["launches"] += 1
        g["ms"] += acc[u] / steps
        g["flops"] += flops
        o = plan.ops[i + cnt - 1]
        per_op.append((name, acc[u] / steps, flops, (o.H, o.Cin, o.Cout, o.k, o.stride)))
    return groups, per_op


def time_postproc(det, plan, x, steps):
    """decode+filter and NMS timed with events (they run on the same stream)."""
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    det.postprocess(plan, x.shape[1])
    torch.cuda.synchronize()
    e[0].record()
    for _ in range(steps):
        det.postprocess(plan, x.shape[1])
    e[1].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / steps


def cpu_baseline(num_classes, side, budget_s=15.0):
    """The oracle on this host's cores: torch-CPU fp32 network + C decode/filter/NMS."""
    from oracle import cref
    from oracle import net_torch as N
    bp, hp = N.calibrated_params(num_classes, seed=0, side=64)
    g = torch.Generator().manual_seed(1)
    batch = 2
    x = torch.rand(batch, side, side, 3, generator=g)

    def step():
        heads = N.detector_forward(bp, hp, x, num_classes)
        heads5 = [h.numpy().reshape(batch, h.shape[1], h.shape[2], 3, 5 + num_classes) for h in heads]
        return cref.bbox_nms(heads5, float(side), num_classes)

    with torch.no_grad():
        step()
        t0 = time.perf_counter()
        n = 0
        while True:
            step()
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= 64:
                break
    return {"value": round(batch * n / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps of batch %d at %dx%d, fp32, torch-CPU convs + C decode/filter/NMS (%.1f s)" %
                      (n, batch, side, side, dt)}


def train_workload(args, dev, rank, world):
    """BASELINE config 5: one training step per timed step -- forward with the graph recorded,
    the YOLO loss at the three scales, back-propagation through every block (input, weight and
    bias gradients), gradient averaging over the ranks (bucketed all-reduces that start
    during the backward pass, N > 1 only), the L2 kernel regulariser's gradient, plain SGD.  Batch = images per GPU (512 / 8 = 64 in the configuration)."""
    import numpy as np
    import torch.distributed as dist
    import yolo_amd as Y
    from yolo_amd import Detector, dataset as DS, losses as L, training as T
    from yolo_amd import dist as ydist
    from yolo_amd.layers import YOLOConv
    B, S, C = args.batch, args.size, args.classes
    cfg = Y.config
    det = Detector(num_classes=C, dtype=torch.bfloat16, seed=0).to(dev)          # same seed: replicated weights
    cal = torch.rand(2, S, S, 3, generator=torch.Generator(device="cpu").manual_seed(100)).to(dev)
    det.calibrate_(cal)
    x = torch.rand(B, S, S, 3, generator=torch.Generator(device="cpu").manual_seed(100 + rank)).to(dev)
    rng = np.random.default_rng(200 + rank)
    nmax = 32
    counts = rng.integers(1, nmax, B).astype(np.int32)
    boxes = np.zeros((B, nmax, 5), np.float32)
    for b in range(B):
        n = counts[b]
        wh = np.exp(rng.normal(3.5, 1.0, (n, 2))).clip(4, S * 0.9)
        ctr = rng.uniform(0, S, (n, 2))
        x0 = np.clip(ctr - wh / 2, 0, S - 2)
        x1 = np.clip(ctr + wh / 2, x0 + 1, S - 1)
        boxes[b, :n] = np.floor(np.concatenate([x0, x1, rng.integers(0, C, (n, 1))], 1))
    targets = DS.pack_targets(*DS.preprocess_true_boxes(torch.from_numpy(boxes).to(dev), torch.from_numpy(counts).to(dev), S, C))
    fns = L.get_yolo_loss_at_scales([0, 1, 2], C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"])   # scales 2, 1, 0
    params = list(det.backbone.parameters()) + list(det.head.parameters())
    buckets = ydist.GradientBuckets(params) if world > 1 else None
    reg = T.L2Regulariser(det)
    lr = 1e-7
    flops = [0.0]

    def count(mod, inp, out):
        flops[0] += 2.0 * out.shape[0] * out.shape[1] * out.shape[2] * mod.kernel_size ** 2 * mod.in_channels * mod.filters

    def step():
        with T.backprop():
            s, m, l = det.head(*det.backbone(x))
        loss = fns[0](targets[2], l) + fns[1](targets[1], m) + fns[2](targets[0], s)
        if buckets is None:
            for p in params:
                p.grad = None
        else:
            buckets.begin()
        loss.backward()                  # with buckets: each bucket's all-reduce starts as its last gradient lands
        if buckets is not None:
            buckets.wait()
        reg.add_gradients_()             # kernel_regularizer l2(0.0005) of every convolution (layers.py:61)
        with torch.no_grad():
            torch._foreach_add_(params, [p.grad for p in params], alpha=-lr)
        return loss

    hooks = [m.register_forward_hook(count) for m in det.modules() if isinstance(m, YOLOConv)]
    first = float(step().detach())
    for h in hooks:
        h.remove()
    for _ in range(max(args.warmup - 1, 0)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    last = float(loss.detach())
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    total_flops = 3.0 * flops[0]                     # forward, input gradient, weight gradient
    result = {
        "metric": "images/sec (training step: fwd + YOLO loss + bwd + SGD) at %d^2" % S,
        "value": round(B * world * args.steps / dt, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "training step: Darknet53+YoloV3(C=%d) forward with saved pre-activations, YOLO loss at three scales, "
                               "backward (dgrad on the forward kernels, MFMA wgrad, bias grad), %sL2 kernel regulariser, SGD; %dx%d, batch %d per GPU, bf16 "
                               "storage / fp32 accumulate and fp32 master weights, activation-calibrated random weights"
                               % (C, "bucketed gradient all-reduce overlapped with the backward pass, " if world > 1 else "", S, S, B),
                   "global_batch": B * world, "per_gpu_batch": B, "image_size": S, "num_classes": C,
                   "parallelism": "dp%d" % world if world > 1 else "single GPU", "loss_first": round(first, 3), "loss_last": round(last, 3),
                   "peak_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)},
        "roofline": {"bound": "mfma", "kernel": "all convolution kernels of the step (forward + input gradient + weight gradient)",
                     "achieved": round(total_flops / (ms * 1e-3) / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(total_flops / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": None,
                     "flops_per_step": total_flops, "measured": "whole step, wall clock (elementwise kernels and the loss included in the time)"},
    }
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_train(C, S)
    return result


def cpu_baseline_train(C, side, budget_s=25.0):
    """The oracle's training step on this host: torch-CPU fp32 forward of the restated network
    with autograd, the restated loss in float64, backward; batch 1."""
    import numpy as np
    from oracle import net_torch as N
    from oracle import loss_torch as LT
    from oracle import postproc_np as P
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    bp = N.init_params(N.darknet53_specs(), seed=0, init="reference")
    hp = N.init_params(N.yolov3_specs(C), seed=1, init="reference")
    for plist in (bp, hp):
        for p in plist:
            for name in ("kernel", "bias"):
                p[name] = p[name].clone().requires_grad_(True)
    x = torch.rand(1, side, side, 3, generator=torch.Generator().manual_seed(1))
    D = 5 + C

    def step():
        heads = N.detector_forward(bp, hp, x, C)
        total = 0
        for i, h in enumerate(heads):
            Si = h.shape[1]
            conv5 = h.double().reshape(1, Si, Si, 3, D)
            label = torch.zeros_like(conv5)
            label[0, Si // 2, Si // 2, 1] = torch.tensor([side / 2, side / 2, 40.0, 40.0, 1.0] + [0.0] * C, dtype=torch.float64)
            bboxes = torch.zeros(1, 128, 4, dtype=torch.float64)
            bboxes[0, 0] = torch.tensor([side / 2, side / 2, 40.0, 40.0], dtype=torch.float64)
            pred = LT.decode_train(conv5, i, Si, P.STRIDES, P.XYSCALE, P.ANCHORS)
            g, c, p = LT.yolo_loss(pred, conv5, label, bboxes, side, 0.5)
            total = total + g + c + p
        total.backward()

    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 8:
            break
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d training steps of batch 1 at %dx%d, fp32 torch-CPU forward + autograd backward, loss in float64 (%.1f s)" % (n, side, side, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the short nms-stress and train legs that the default single-GPU run appends under breakdown")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="the K timed steps are repeated until the timed region lasts about this long (0: exactly K steps)")
    ap.add_argument("--per-op", action="store_true", help="also print the per-launch table to stderr")
    ap.add_argument("--workload", default="detector", choices=["detector", "nms-stress", "train"],
                    help="detector (default): the whole hot path; nms-stress: decode+filter+NMS alone on 100 800 "
                         "synthetic candidates per image (use --batch 32); train: one training step per step "
                         "(forward, YOLO loss, backward, SGD; gradient all-reduce when --gpus > 1)")
    ap.add_argument("--split", default="auto", help="sub-batches run concurrently on separate streams: "
                    "auto (timed once), off, or sizes such as 48,16")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # Rehearsal knobs for boxes with fewer GPUs than ranks (not used by the driver): every rank
    # on one device, gloo instead of RCCL.
    backend = os.environ.get("YMX_BENCH_BACKEND", "nccl")
    if os.environ.get("YMX_BENCH_DEVICE") is not None:
        local = int(os.environ["YMX_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.workload == "nms-stress":
        if world != 1:
            raise SystemExit("the nms-stress workload is a single-GPU measurement")
        if args.batch == 64:
            args.batch = 32                          # the configuration's batch unless overridden
        print(json.dumps(stress_workload(args, dev)), flush=True)
        return
    if args.workload == "train":
        if args.steps == 20:
            args.steps = 10
        result = train_workload(args, dev, rank, world)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    from yolo_amd import Detector
    from yolo_amd import dist as ydist

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B, S, C = args.batch, args.size, args.classes
    split = args.split if args.split in ("auto", "off") else tuple(int(v) for v in args.split.split(","))
    det = Detector(num_classes=C, dtype=dtype, seed=0, split=split).to(dev)
    gen = torch.Generator(device="cpu").manual_seed(100 + rank)
    x = torch.rand(B, S, S, 3, generator=gen).to(dev)           # synthetic images, resident in HBM
    det.calibrate_(x[:2].contiguous())
    global_batch = B * world

    def step():
        out = det(x)
        if world > 1:
            out = ydist.all_gather_detections(out, global_batch)
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # The timed region is `repeats` back-to-back passes of the K steps, sized from one untimed
    # step so that it lasts about two seconds (K steps alone are ~0.2 s: too short for a
    # sampling power / activity monitor to see); every figure below is per step.
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    est = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([est], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        est = float(t.item())
    repeats = 1 if args.min_seconds <= 0 else max(1, min(200, int(-(-args.min_seconds // max(est * args.steps, 1e-6)))))
    timed_steps = args.steps * repeats
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(timed_steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / timed_steps * 1e3
    value = global_batch * timed_steps / elapsed
    valid_mean = float(out[3].float().mean().item())

    line = {
        "metric": "images/sec (fwd+decode+NMS) at %d^2" % S,
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic", "repeats": repeats, "timed_steps": timed_steps,
        "timed_seconds": round(elapsed, 3),
        "config": {"workload": "Darknet53+YoloV3(C=%d) forward + decode + filter + combined NMS, %dx%d, batch %d per GPU, "
                               "%s storage / fp32 accumulate, random weights (activation-calibrated)" % (C, S, S, B, args.dtype),
                   "global_batch": global_batch, "per_gpu_batch": B, "image_size": S, "num_classes": C,
                   "parallelism": "batch shard x%d + all-gather of detections" % world if world > 1 else "single GPU",
                   "mean_valid_detections": valid_mean},
    }

    line["config"]["sub_batches"] = list(det._plan(B, S, x.dtype, dev).sizes)

    if rank == 0 and not args.no_profile:
        # Kernel-level figures come from a separate pass over the whole batch as ONE program,
        # every launch alone on the GPU (the timed region above may run the batch as
        # concurrent sub-batches, where per-launch durations overlap and mean nothing).
        whole = det._build_plan((B,), S, x.dtype, dev)
        groups, per_op = profile_ops(whole, x, max(3, min(args.steps, 10)))
        dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
        name, g = dom
        achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
        conv_ms = sum(v["ms"] for v in groups.values())
        conv_flops_total = sum(v["flops"] for v in groups.values())
        line["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(name),
                            "launches_per_step": g["launches"], "avg_launch_ms": round(g["ms"] / g["launches"], 4),
                            "flops_per_step": g["flops"], "launch_batch": B,
                            "measured": "whole batch as one program, each launch alone on the GPU"}
        plan = whole
        post_ms = time_postproc(det, plan, x, max(3, min(args.steps, 10)))
        part = plan.parts[0]
        dec_ms, nms_ms = time_postproc_split(lambda: det._decode_part(part, S), lambda: det._nms_part(part),
                                             max(3, min(args.steps, 10)))
        N = plan.scales.N
        m_mean = float(plan.counts.float().mean().item())
        dec_bytes, nms_bytes = postproc_bytes(B, N, C, m_mean, 4)
        # the bandwidth-bound half of the path, on the same line the driver keeps
        line["roofline"]["hbm"] = {
            "decode": hbm_roofline("decode_kernel<float, 20, 1, true>", dec_bytes, dec_ms,
                                   pmc_traffic(None, kernel="decode_kernel<float, 20, 1, true>")),
            "nms": hbm_roofline("nms_class_kernel + nms_merge_kernel", nms_bytes, nms_ms,
                                _sum_traffic("r*_pmc_traffic.json", ("nms_",)))}
        line["breakdown"] = {
            "conv_stack_ms": round(conv_ms, 4), "conv_stack_tflops": round(conv_flops_total / (conv_ms * 1e-3) / 1e12, 2),
            "conv_stack_frac_of_bf16_peak": round(conv_flops_total / (conv_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "postproc_ms": round(post_ms, 4), "postproc_algorithmic_GBs": round((dec_bytes + nms_bytes) / (post_ms * 1e-3) / 1e9, 2),
            "postproc_frac_of_hbm_peak": round((dec_bytes + nms_bytes) / (post_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "mean_survivors_per_image": m_mean,
            "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 4),
                            "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None}
                        for k, v in groups.items()}}
        if args.per_op:
            for i, (nm, ms, fl, shp) in enumerate(per_op):
                sys.stderr.write("%3d %-28s HxW=%-4d Cin=%-5d Cout=%-5d k=%d s=%d  %8.4f ms  %8.1f TF\n" %
                                 (i, nm, shp[0], shp[1], shp[2], shp[3], shp[4], ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(C, S)
    if rank == 0 and world == 1 and not args.no_extra_legs and S == 416 and B == 64:
        # BASELINE configs 4 and 5 on the same line, each a short leg of its own workload with its
        # own roofline and CPU baseline
        sub = argparse.Namespace(**vars(args))
        sub.batch, sub.steps, sub.warmup = 32, 20, 3
        leg = stress_workload(sub, dev)
        line.setdefault("breakdown", {})["stress"] = {k: leg[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype",
                                                                           "candidates_per_s", "config", "roofline", "roofline_nms",
                                                                           "cpu_baseline") if k in leg}
        sub.batch, sub.steps, sub.warmup = 64, 6, 2
        leg = train_workload(sub, dev, 0, 1)
        line["breakdown"]["train"] = {k: leg[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config",
                                                           "roofline", "cpu_baseline") if k in leg}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
