"""The RCCL transport on the one GPU a test box has: a process group of a single rank over the
"nccl" backend (RCCL on ROCm).  Every collective is then an identity, but the calls go through
RCCL with device buffers, which is the code path `bench.py --gpus N` takes on a multi-GPU node
and which the gloo rehearsals in test_host.py replace by host copies."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
import yolo_amd
from yolo_amd import dist as D
from yolo_amd.utils import Detections

ok = True
# 1. detections of a real forward pass (detector + decode + filter + NMS), gathered on the device
det = yolo_amd.Detector(num_classes=3, dtype=torch.bfloat16, seed=0, split="off").to(dev)
x = torch.rand(4, 64, 64, 3, generator=torch.Generator().manual_seed(1)).to(dev)
det.calibrate_(x[:2].contiguous())
out = det(x)
got = D.all_gather_detections(out, 4)
torch.cuda.synchronize()
ok = ok and got[0].is_cuda and all(torch.equal(a, b) for a, b in zip(tuple(out) + (out.keep_idx,), tuple(got) + (got.keep_idx,)))
# a shard size check still applies in a world of one
try:
    D.all_gather_detections(out, 5)
    ok = False
except ValueError:
    pass
# 2. gradient buckets on device buffers: asynchronous all-reduces with ReduceOp.AVG, started from
#    autograd's hooks while backward() runs; in a group of one the average is the gradient itself
torch.manual_seed(3)
net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40), torch.nn.Tanh(),
                          torch.nn.Linear(40, 2)).to(dev)
nb = D.GradientBuckets(list(net.parameters()), bucket_bytes=2048, rehearse=True)
xs = torch.randn(5, 6, generator=torch.Generator().manual_seed(11)).to(dev)
for step in range(2):
    nb.begin()
    net(xs).square().sum().backward()
    ok = ok and len(nb.buckets) >= 3 and nb.launch_order == list(range(len(nb.buckets))) and len(nb._handles) == len(nb.buckets)
    nb.wait()
    ref = torch.autograd.grad(net(xs).square().sum(), list(net.parameters()))
    ok = ok and all(p.grad.is_cuda and torch.allclose(p.grad, g, rtol=1e-5, atol=1e-6) for p, g in zip(net.parameters(), ref))
nb.close()
# 3. the timing exchange of bench.py (MAX over ranks of a device scalar) and its barrier
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
ok = ok and float(t.item()) == 1.25
dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


@pytest.mark.gpu
def test_rccl_single_rank_gather_and_buckets(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_worker.py"
    script.write_text(_WORKER % {"root": ROOT, "port": port})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
