# Samples socket power and clocks while the detector step runs in a loop (diagnostic).
set -e
python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from yolo_amd import Detector
dev = torch.device("cuda", 0)
det = Detector(num_classes=80, dtype=torch.bfloat16, seed=0, split="off").to(dev)
x = torch.rand(64, 416, 416, 3, generator=torch.Generator().manual_seed(100)).to(dev)
det.calibrate_(x[:2].contiguous())
t_end = time.time() + 14
n = 0
torch.cuda.synchronize(); t0 = time.time()
while time.time() < t_end:
    for _ in range(20):
        det(x)
    torch.cuda.synchronize(); n += 20
print("steps %d  ms/step %.3f" % (n, (time.time() - t0) / n * 1e3), flush=True)
PY
PID=$!
sleep 9
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)|Performance Level" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $PID
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" || true
