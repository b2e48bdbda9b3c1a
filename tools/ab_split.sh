set -e
for s in off 48,16 40,24 32,32 56,8 off 48,16; do
  echo "split $s"; timeout -k 10 120 python bench.py --no-profile --no-extra-legs --no-cpu-baseline --split $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['sub_batches'])"
done
timeout -k 10 200 python tools/conv_bench.py 64,13,13,1024,512,1,1 64,52,52,256,128,1,1 64,26,26,512,256,1,1 64,13,13,1024,256,1,1
