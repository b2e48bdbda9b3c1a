"""Batch-sharded multi-GPU execution: one process per GPU; inference needs one all-gather
at the end, a training step one bucketed all-reduce of the gradients (GradientBuckets).

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


class GradientBuckets(object):
    """Data-parallel gradient averaging for a training step (the exchange a batch-sharded
    training run needs; the reference has no multi-device code).

    The gradients of ``params`` live in a few large flat buffers (``bucket_bytes`` each,
    filled in reverse parameter order, the order back-propagation produces them);
    ``p.grad`` of every parameter is a view into its bucket, so nothing is copied before or
    after the collective.  ``reduce()`` starts one asynchronous all-reduce per bucket and
    ``wait()`` finishes them and divides by the world size.  On RCCL a ring all-reduce moves
    2 (n-1)/n of a bucket over each GPU's xGMI links; 64 MiB buckets keep the six collectives
    of an 88 M-parameter fp32 model bandwidth-bound (about 0.5 ms each at 150 GB/s per link)
    while leaving the first ones free to overlap with the rest of the backward pass."""

    def __init__(self, params, bucket_bytes=64 << 20, group=None):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []                       # flat tensors
        self._handles = []
        order = list(reversed(self.params))
        i = 0
        while i < len(order):
            dtype, device = order[i].dtype, order[i].device
            chunk, size = [], 0
            while i < len(order) and order[i].dtype == dtype and order[i].device == device and \
                    (not chunk or (size + order[i].numel()) * order[i].element_size() <= bucket_bytes):
                chunk.append(order[i])
                size += order[i].numel()
                i += 1
            flat = torch.zeros((size,), dtype=dtype, device=device)
            off = 0
            for p in chunk:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.buckets.append(flat)

    def zero_(self):
        for b in self.buckets:
            b.zero_()

    def reduce(self):
        """Start the all-reduce of every bucket (in the order the backward pass fills them)."""
        if not (dist.is_available() and dist.is_initialized()):
            return self
        for b in self.buckets:
            t = b
            if b.is_cuda and dist.get_backend(self.group) != "nccl":
                t = b.cpu()                     # rehearsal backends reduce host tensors
            self._handles.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), t, b))
        return self

    def wait(self):
        """Finish the collectives; gradients become the average over the ranks."""
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        for h, t, b in self._handles:
            h.wait()
            if t is not b:
                b.copy_(t)
            if world > 1:
                b.div_(world)
        self._handles = []
        return self
