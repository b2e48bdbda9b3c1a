"""Batch-sharded multi-GPU inference: one process per GPU, one all-gather at the end.

The reference has no multi-device code.  Images are independent through conv, decode,
filter and NMS, so the global batch is split contiguously by rank, every rank runs the
whole detector on its shard with replicated weights, and the only exchange is an
all-gather of the fixed-size detection records (``max_outputs`` rows per image), which
over RCCL is one small, latency-bound collective on the xGMI mesh.
"""
import torch
import torch.distributed as dist

from .utils import Detections


def shard_range(global_batch, rank, world_size):
    """Contiguous [start, start+count) of ``global_batch`` images owned by ``rank``;
    the first ``global_batch % world_size`` ranks hold one extra image."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def pack_detections(det, rows):
    """Detections of b images -> one float32 tensor [rows, T*7+1] (int fields bit-cast),
    zero/-1 padded to ``rows`` images so every rank contributes the same size."""
    boxes, scores, classes, valid = det
    keep = det.keep_idx
    b, t = scores.shape
    rec = torch.cat([boxes.reshape(b, 4 * t), scores, classes, keep.view(torch.float32),
                     valid.view(torch.float32).reshape(b, 1)], dim=1)        # one kernel
    if rows > b:                                                             # uneven shards only
        pad = torch.zeros((rows - b, t * 7 + 1), device=rec.device, dtype=torch.float32)
        pad[:, 6 * t:7 * t] = torch.full((rows - b, t), -1, device=rec.device, dtype=torch.int32).view(torch.float32)
        rec = torch.cat([rec, pad], dim=0)
    return rec


def unpack_detections(rec, t):
    n = rec.shape[0]
    boxes = rec[:, 0:4 * t].reshape(n, t, 4).contiguous()
    scores = rec[:, 4 * t:5 * t].contiguous()
    classes = rec[:, 5 * t:6 * t].contiguous()
    keep = rec[:, 6 * t:7 * t].contiguous().view(torch.int32)
    valid = rec[:, 7 * t].contiguous().view(torch.int32)
    return Detections(boxes, scores, classes, valid, keep)


def all_gather_detections(det, global_batch, group=None):
    """Gather every rank's shard of detections into the global-batch order.

    ``det`` holds this rank's shard (``shard_range`` of ``global_batch``).  Returns
    Detections for all ``global_batch`` images on every rank; bit-identical to running
    the whole batch on one GPU because each image's result depends on that image alone.
    """
    if not dist.is_available() or not dist.is_initialized():
        return det
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    t = det[1].shape[1]
    rows = -(-global_batch // world)
    _, mine = shard_range(global_batch, rank, world)
    if det[1].shape[0] != mine:
        raise ValueError("rank %d holds %d images, its shard is %d" % (rank, det[1].shape[0], mine))
    rec = pack_detections(det, rows)
    device = rec.device
    if rec.is_cuda and dist.get_backend(group) != "nccl":
        rec = rec.cpu()             # gloo gathers host tensors only; RCCL takes the device buffer as is
    out = torch.empty((world * rows, rec.shape[1]), device=rec.device, dtype=rec.dtype)
    dist.all_gather_into_tensor(out, rec, group=group)           # one collective, one output buffer
    if out.device != device:
        out = out.to(device)
    if global_batch == world * rows:                             # even shards: the buffer is already in order
        return unpack_detections(out, t)
    out = out.view(world, rows, -1)
    parts = []
    for r in range(world):
        _, cnt = shard_range(global_batch, r, world)
        parts.append(out[r, :cnt])
    return unpack_detections(torch.cat(parts, dim=0), t)
