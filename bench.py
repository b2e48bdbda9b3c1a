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
                host on a bounded sample of the same workload (rank 0, N = 1 only).
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
    "conv_igemm_halo_bf16_256x256": "conv_igemm_halo_kernel<bf16>",
    "conv_igemm_dma_bf16_256x256": "conv_igemm_dma_kernel<bf16, 256>",
    "conv_igemm_dma4_bf16_256x256": "conv_igemm_dma4_kernel<bf16>",
}


def pmc_traffic(label):
    """HBM bytes per launch of the labelled kernel, from the newest committed --pmc summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files or label not in PMC_KERNEL:
        return None
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"].get(PMC_KERNEL[label])
        return None if k is None else round(k["hbm_bytes_per_launch"], 1)
    except (OSError, ValueError, KeyError):
        return None


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
    n = plan.n_ops
    acc = [0.0] * n
    elems = [plan.ops[i] for i in range(n)]                      # views into the program array
    op_ptr = [_lib.ctypes.byref(e) for e in elems]
    for _ in range(steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        for i in range(n):
            ev[i].record()
            _lib.check(L.ymx_run_program(op_ptr[i], 1, _lib.stream()))
        ev[n].record()
        torch.cuda.synchronize()
        for i in range(n):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
    groups = {}
    for i in range(n):
        name = conv_kernel_name(plan.ops[i])
        g = groups.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
        g["launches"] += 1
        g["ms"] += acc[i] / steps
        g["flops"] += conv_flops(plan.ops[i])
    per_op = [(conv_kernel_name(plan.ops[i]), acc[i] / steps, conv_flops(plan.ops[i]),
               (plan.ops[i].H, plan.ops[i].Cin, plan.ops[i].Cout, plan.ops[i].k, plan.ops[i].stride)) for i in range(n)]
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
    ap.add_argument("--per-op", action="store_true", help="also print the per-launch table to stderr")
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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
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
    ms_per_step = elapsed / args.steps * 1e3
    value = global_batch * args.steps / elapsed
    valid_mean = float(out[3].float().mean().item())

    line = {
        "metric": "images/sec (fwd+decode+NMS) at 416^2",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
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
        N = plan.scales.N
        m_mean = float(plan.counts.float().mean().item())
        head_e = 4
        dec_bytes = B * (N * (5 + C) * head_e + m_mean * (4 + C + 1) * 4 + 4)
        nms_bytes = B * (m_mean * (4 + C) * 4 + 32 * 6 * 4 + 4 + 32 * 4)
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
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
