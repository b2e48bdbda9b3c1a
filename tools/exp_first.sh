timeout -k 10 300 python3 -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q --timeout 300 --timeout-method thread -k "conv_goldens or conv_random" | tail -2
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --per-op 2>&1 >/dev/null | head -5
