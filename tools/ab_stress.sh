#!/bin/bash
# A/B of the stress bench: product library vs YMX_LIBRARY=$1 (same box, alternating)
set -e
O=$PWD/gpurun_out/ab; rm -rf $O; mkdir -p $O
for i in 1 2; do
  timeout -k 10 200 python3 bench.py --workload nms-stress --steps 20 --warmup 3 --no-cpu-baseline > $O/new_$i.json
  YMX_LIBRARY=$1 timeout -k 10 200 python3 bench.py --workload nms-stress --steps 20 --warmup 3 --no-cpu-baseline > $O/base_$i.json
done
python3 - <<PY
import json
for n in ("new_1","base_1","new_2","base_2"):
    d=json.load(open("$O/%s.json"%n))
    print(n,'step ms', d['ms_per_step'], 'decode', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'nms', d['roofline_nms']['avg_launch_ms'], d['roofline_nms']['frac'])
PY
