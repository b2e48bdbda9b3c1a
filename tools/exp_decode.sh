for r in 1 2; do for v in N7 N6 N4; do
  echo -n "$v "; YMX_LIBRARY=$PWD/yolo_amd/libymx_exp_$v.so timeout -k 10 200 python3 bench.py --workload nms-stress --steps 30 --warmup 3 --no-cpu-baseline | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nms ms', d['roofline_nms']['avg_launch_ms'])"
done; done
