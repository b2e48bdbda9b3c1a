#!/bin/bash
# All rocprofv3 evidence of one round (run on the GPU box from the repo root):
#   bash tools/prof_round.sh r02
# Writes summaries to gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
set -e
TAG=${1:-rXX}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
BARGS="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-extra-legs --min-seconds 0 --split off"
PARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extra-legs --min-seconds 0 --split off"
SARGS="bench.py --workload nms-stress --steps 10 --warmup 2 --no-cpu-baseline"

# detector: kernel trace of the serial program, then the default (concurrent sub-batches)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_off -- python3 $BARGS > $O/t_off.log 2>&1
python3 tools/prof_summary.py $(find $O/t_off -name '*kernel_trace.csv' | head -1) 20 1 > $O/${TAG}_bench_kernel_trace_summary.txt
cp $(find $O/t_off -name '*kernel_stats.csv' | head -1) $O/${TAG}_bench_kernel_stats_raw.csv
# detector: HBM traffic, two separate passes
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python3 $PARGS > $O/p_fetch.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python3 $PARGS > $O/p_write.log 2>&1
python3 tools/pmc_summary.py $(find $O/p_fetch -name '*counter_collection.csv' | head -1) $(find $O/p_write -name '*counter_collection.csv' | head -1) 1 > $O/${TAG}_pmc_traffic.json
# detector: MFMA utilisation
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p_mfma -- python3 $PARGS > $O/p_mfma.log 2>&1
python3 tools/pmc_mfma_summary.py $(find $O/p_mfma -name '*counter_collection.csv' | head -1) 1 > $O/${TAG}_pmc_mfma.json
# NMS stress: trace + traffic
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_trace -- python3 $SARGS > $O/s_trace.log 2>&1
cp $(find $O/s_trace -name '*kernel_stats.csv' | head -1) $O/${TAG}_stress_kernel_stats_raw.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/s_fetch -- python3 $SARGS > $O/s_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/s_write -- python3 $SARGS > $O/s_write.log 2>&1
python3 tools/pmc_summary.py $(find $O/s_fetch -name '*counter_collection.csv' | head -1) $(find $O/s_write -name '*counter_collection.csv' | head -1) 2 > $O/${TAG}_pmc_traffic_stress.json
# training step: kernel trace (everything on one stream, so per-kernel times add up), then the same step without the profiler
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/tr -o tr -- python3 tools/train_step_time.py 64 416 4 > $O/tr.log 2>&1
timeout -k 10 200 python3 tools/train_step_time.py 64 416 6 > $O/tr_plain.log 2>&1
(echo "rocprofv3 --kernel-trace of tools/train_step_time.py 64 416 4 (one MI355X; 5 steps incl. the first); without the profiler: $(grep 'train step' $O/tr_plain.log | tail -1)"; echo "columns: kernel, launches per step, ms per step"; python3 tools/trace_summary.py $(find $O/tr -name 'tr_results.db' | head -1) 5) > $O/${TAG}_train_kernel_trace_summary.txt
# drop the bulky raw outputs, keep the summaries
rm -rf $O/t_off $O/p_fetch $O/p_write $O/p_mfma $O/s_trace $O/s_fetch $O/s_write $O/tr
ls -la $O
head -14 $O/${TAG}_bench_kernel_trace_summary.txt
