"""Per-kernel summary of a rocprofv3 --kernel-trace CSV for bench.py's timed steps.

    python tools/prof_summary.py <..._kernel_trace.csv> [timed_steps [sub_batches]]

bench.py first calibrates the random weights (about 150 small launches at batch 2), times
the candidate cuts of the batch and runs its warm-up steps; those would distort per-kernel
averages, so only the dispatches of the last `timed_steps` steps are counted (run bench.py
with --no-profile so that the trace ends with them).  Every sub-batch of a step ends with
one nms_merge_kernel dispatch, which is how steps are told apart (sub_batches = 2 when the
engine runs the batch as two concurrent sub-batches).  With concurrent sub-batches the
per-kernel durations overlap: their sum exceeds the wall span.
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
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    parts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    merges = [i for i, r in enumerate(rows) if "nms_merge_kernel" in r["Kernel_Name"]]
    need = steps * parts
    if len(merges) < need + 1:
        raise SystemExit("trace holds %d nms_merge dispatches, %d steps x %d need more" % (len(merges), steps, parts))
    first = merges[-need - 1] + 1
    rows = rows[first:merges[-1] + 1]
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += d
    total = sum(v[1] for v in agg.values())
    span = max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])
    print("steps counted: %d   sub-batches per step: %d   kernel time per step: %.3f ms   wall span per step: %.3f ms" %
          (steps, parts, total / steps / 1e6, span / steps / 1e6))
    print("%-62s %9s %12s %10s %7s" % ("kernel", "calls/step", "avg us", "ms/step", "%"))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-62s %9.1f %12.2f %10.4f %6.2f%%" % (k[:62], n / steps, t / n / 1e3, t / steps / 1e6, 100.0 * t / total))


if __name__ == "__main__":
    main()
