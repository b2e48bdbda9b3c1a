#!/bin/bash
# Post-processing tests, then the NMS-stress bench with a kernel trace (GPU box, repo root).
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/qs
rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_kernels.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread -k "decode or filter or nms or alloc" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 bench.py --workload nms-stress --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json'))
print('step ms', d['ms_per_step'], 'decode', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'nms', d['roofline_nms']['avg_launch_ms'], d['roofline_nms']['frac'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --workload nms-stress --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
S=$(find $O/trace -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$S")))[:4]:
    print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $O/trace
