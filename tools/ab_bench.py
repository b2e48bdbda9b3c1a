"""Runs bench.py against two builds of the library in alternation (tuning aid).

    python tools/ab_bench.py libymx.so libymx_exp_x.so [rounds]

Each run is a fresh process (the library path is fixed at first use); prints ms per step.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import os, sys
sys.path.insert(0, %r)
from yolo_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), %r)
sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--no-profile', '--no-extra-legs']
import runpy
runpy.run_path(os.path.join(%r, 'bench.py'), run_name='__main__')
"""


def main():
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for r in range(rounds):
        for lib in libs:
            out = subprocess.run([sys.executable, "-c", CODE % (ROOT, lib, ROOT)], capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(lib, "FAILED", out.stderr[-400:])
                continue
            d = json.loads(line[-1])
            print("%-24s %.3f ms/step  %.0f img/s" % (lib, d["ms_per_step"], d["value"]), flush=True)


if __name__ == "__main__":
    main()
