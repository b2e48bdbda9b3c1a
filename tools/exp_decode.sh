for r in 1 2; do for v in V0 V1 V2 V3 V4; do
  echo -n "$v "; YMX_LIBRARY=$PWD/yolo_amd/libymx_exp_$v.so timeout -k 10 200 python3 bench.py --workload nms-stress --steps 30 --warmup 3 --no-cpu-baseline | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('decode ms', d['roofline']['avg_launch_ms'])"
done; done
