"""Prints the interesting fields of a bench.py JSON line (last line of the given file)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f img/s  ms/step %.3f  dominant %s %.1f TF (%.3f)  conv %.3f ms  postproc %.3f ms" % (
    d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"],
    d["breakdown"]["conv_stack_ms"], d["breakdown"]["postproc_ms"]))
for k, v in d["breakdown"]["kernels"].items():
    print("  %-28s launches %3d  %8.3f ms  %s TF" % (k, v["launches"], v["ms"], v["tflops"]))
