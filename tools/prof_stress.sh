#!/bin/bash
# Kernel trace + HBM traffic passes of the NMS-stress workload (run on the GPU box from the repo root).
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/stress
rm -rf $O; mkdir -p $O
ARGS="bench.py --workload nms-stress --steps 10 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 python3 bench.py --workload nms-stress --steps 20 --warmup 3 > $O/bench.json
cat $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ARGS > $O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $ARGS > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $ARGS > $O/write.log 2>&1
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)
W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $F $W 2 > $O/pmc_traffic_stress.json
S=$(find $O/trace -name '*kernel_stats.csv' | head -1)
cp $S $O/kernel_stats.csv
head -12 $O/kernel_stats.csv
# keep the merged output small
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete
