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
    training run needs; the reference has no multi-device code), overlapped with the
    backward pass.

    The gradients of ``params`` live in a few large flat buffers (``bucket_bytes`` each,
    filled in reverse parameter order, the order back-propagation produces them; every
    parameter starts on a 256-byte boundary); ``p.grad`` of every parameter is a view into
    its bucket, so nothing is copied before or after the collective.  A bucket's
    asynchronous all-reduce starts the moment its last gradient is final, while the rest of
    the backward pass still runs:

      * yolo_amd.training's convolution blocks write their weight and bias gradients
        straight into the views (``p._grad_bucket``) and call ``ready(p)``;
      * for any other parameter a post-accumulate-grad hook calls ``ready(p)`` once autograd
        has added its gradient into the view.

    Buckets start in index order (a bucket whose turn has not come waits for the ones before
    it), so every rank issues the same collectives in the same order whatever order its
    gradients arrive in.  Per step::

        buckets.begin()          # zero the buffers, forget last step's state
        loss.backward()          # all-reduces start as buckets fill
        buckets.wait()           # starts what is left, finishes all; .grad = mean over ranks

    On RCCL a ring all-reduce moves 2 (n-1)/n of a bucket over each GPU's xGMI links; 64 MiB
    buckets keep the six collectives of an 88 M-parameter fp32 model bandwidth-bound (about
    0.5 ms each at 150 GB/s per link) and the first five hide behind the backward pass."""

    ALIGN = 64                                  # elements: 256 bytes of fp32

    def __init__(self, params, bucket_bytes=64 << 20, group=None, hooks=True, rehearse=False):
        self.group = group
        self.rehearse = bool(rehearse)          # issue the collectives in a group of one rank too (each is then an identity):
                                                # runs the transport's code path on a single GPU
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []                       # flat tensors
        self.members = []                       # per bucket: its parameters
        self.launch_order = []                  # bucket indices in the order their collectives started (this step)
        self._where = {}                        # id(param) -> bucket index
        self._pending = []                      # per bucket: ids of parameters not ready yet
        self._handles = {}                      # bucket index -> (work, reduced tensor)
        self._next = 0                          # first bucket not started yet
        self._hooks = []
        order = list(reversed(self.params))
        i = 0
        while i < len(order):
            dtype, device = order[i].dtype, order[i].device
            chunk, offs, size = [], [], 0
            while i < len(order) and order[i].dtype == dtype and order[i].device == device and \
                    (not chunk or (size + order[i].numel()) * order[i].element_size() <= bucket_bytes):
                chunk.append(order[i])
                offs.append(size)
                size += -(-order[i].numel() // self.ALIGN) * self.ALIGN
                i += 1
            flat = torch.zeros((size,), dtype=dtype, device=device)
            for p, off in zip(chunk, offs):
                view = flat[off:off + p.numel()].view_as(p)
                p.grad = view
                p._grad_bucket = (self, view)
                self._where[id(p)] = len(self.buckets)
            self.buckets.append(flat)
            self.members.append(chunk)
        if hooks:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self.ready))
        self.begin(zero=False)

    def _active(self):
        return dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size(self.group) > 1 or self.rehearse)

    def begin(self, zero=True):
        """Start of a step: gradients zeroed (autograd adds into them), nothing ready."""
        if self._handles:
            raise RuntimeError("GradientBuckets.begin: the previous step's collectives were not waited for")
        if zero:
            for b in self.buckets:
                b.zero_()
        self._pending = [set(id(p) for p in chunk) for chunk in self.members]
        self._next = 0
        self.launch_order = []
        return self

    zero_ = begin

    def _start(self, i):
        b = self.buckets[i]
        self.launch_order.append(i)
        if not self._active():
            return
        t = b
        nccl = dist.get_backend(self.group) == "nccl"
        if b.is_cuda and not nccl:
            t = b.cpu()                         # rehearsal backends reduce host tensors
        op = dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM
        self._handles[i] = (dist.all_reduce(t, op=op, group=self.group, async_op=True), t, nccl)

    def ready(self, p):
        """The gradient of ``p`` is final for this step (safe to call from the autograd thread)."""
        i = self._where.get(id(p))
        if i is None:
            raise ValueError("GradientBuckets.ready: not one of this object's parameters")
        self._pending[i].discard(id(p))
        while self._next < len(self.buckets) and not self._pending[self._next]:
            self._start(self._next)
            self._next += 1

    def reduce(self):
        """Start the all-reduce of every bucket that has not started yet (parameters that took
        no part in the step, or a backward pass that does not report readiness)."""
        while self._next < len(self.buckets):
            self._start(self._next)
            self._next += 1
        return self

    def wait(self):
        """Finish the collectives; gradients become the average over the ranks."""
        self.reduce()
        world = dist.get_world_size(self.group) if self._active() else 1
        for i in sorted(self._handles):
            h, t, averaged = self._handles[i]
            h.wait()
            b = self.buckets[i]
            if t is not b:
                b.copy_(t)
            if world > 1 and not averaged:
                b.div_(world)
        self._handles = {}
        return self

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def close(self):
        """Detach from the parameters: hooks removed, gradients handed back as ordinary tensors
        (copies of the current values)."""
        self.remove_hooks()
        for p in self.params:
            if getattr(p, "_grad_bucket", (None,))[0] is self:
                del p._grad_bucket
                if p.grad is not None:
                    p.grad = p.grad.clone()
