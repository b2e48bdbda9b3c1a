# A/B of library builds on the stress workload in one box: bash tools/exp_ab.sh <alt .so> [<alt .so> ...]
for i in 1 2; do
for lib in "" "$@"; do
  YMX_LIBRARY=$lib timeout -k 10 120 python3 bench.py --workload nms-stress --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-default}', 'step', d['ms_per_step'], 'decode', d['roofline']['avg_launch_ms'], 'nms', d['roofline_nms']['avg_launch_ms'], d['roofline_nms']['frac'])
" || exit 1
done
done
