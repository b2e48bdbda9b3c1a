"""Per-kernel summary of a rocprofv3 --kernel-trace CSV for bench.py's timed steps.

    python tools/prof_summary.py <..._kernel_trace.csv> [warmup_steps]

bench.py first calibrates the random weights (about 150 small launches at batch 2) and
runs its warm-up steps; those would distort per-kernel averages, so only dispatches after
the warm-up are counted.  A step ends with exactly one nms_merge_kernel dispatch, which
is how steps are told apart.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ymx::", "").replace("ymx::", "")
    name = name.replace("unsigned short", "bf16")
    return name


def main():
    path = sys.argv[1]
    warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    merges = [i for i, r in enumerate(rows) if "nms_merge_kernel" in r["Kernel_Name"]]
    if len(merges) <= warmup:
        raise SystemExit("trace holds %d steps, warm-up is %d" % (len(merges), warmup))
    first = merges[warmup - 1] + 1 if warmup > 0 else 0
    steps = len(merges) - warmup
    rows = rows[first:merges[-1] + 1]
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += d
    total = sum(v[1] for v in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print("steps counted: %d   kernel time per step: %.3f ms   wall span per step: %.3f ms" %
          (steps, total / steps / 1e6, span / steps / 1e6))
    print("%-62s %9s %12s %10s %7s" % ("kernel", "calls/step", "avg us", "ms/step", "%"))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-62s %9.1f %12.2f %10.4f %6.2f%%" % (k[:62], n / steps, t / n / 1e3, t / steps / 1e6, 100.0 * t / total))


if __name__ == "__main__":
    main()
