import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import os, sys
sys.path.insert(0, %r)
from yolo_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), %r)
sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--no-extra-legs']
import runpy
runpy.run_path(os.path.join(%r, 'bench.py'), run_name='__main__')
"""
res = {}
for lib in sys.argv[1:3]:
    out = subprocess.run([sys.executable, "-c", CODE % (ROOT, lib, ROOT)], capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(lib, "FAILED", out.stderr[-800:]); continue
    res[lib] = json.loads(line[-1])
a, b = [res[l] for l in sys.argv[1:3]]
print("step ms", a["ms_per_step"], b["ms_per_step"], "conv stack ms", a["breakdown"]["conv_stack_ms"], b["breakdown"]["conv_stack_ms"])
ka, kb = a["breakdown"]["kernels"], b["breakdown"]["kernels"]
for k in sorted(set(ka) | set(kb)):
    x, y = ka.get(k), kb.get(k)
    print("%-36s %s | %s" % (k, "%2d x  %.4f ms %s TF" % (x["launches"], x["ms"], x["tflops"]) if x else "-", "%2d x  %.4f ms %s TF" % (y["launches"], y["ms"], y["tflops"]) if y else "-"))
print("roofline", a["roofline"]["frac"], b["roofline"]["frac"])
