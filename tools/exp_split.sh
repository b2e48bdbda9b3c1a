# one process per cut, same box: step time of the default workload for several sub-batch cuts
for sp in auto 48,16 32,32 32,16,16 40,24 24,24,16 16,16,16,16 56,8; do
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --split $sp 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$sp', d['ms_per_step'], d['value'], d['config'].get('sub_batches'))
" || exit 1
done
