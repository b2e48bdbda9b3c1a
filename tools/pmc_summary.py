"""HBM traffic per kernel launch from two rocprofv3 --pmc passes over bench.py.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [warmup_steps] > out.json

FETCH_SIZE and WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE tallies the
128-byte requests of wide streaming reads at 64 bytes, so it is doubled here; WRITE_SIZE is
exact for 16-byte-per-lane stores.  Only dispatches after the warm-up steps are counted (a
step ends with one nms_merge_kernel dispatch; the weight-calibration launches before the
first step are skipped the same way).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ymx::", "").replace("ymx::", "")
    return name.replace("unsigned short", "bf16")


def per_kernel(path, counter, warmup):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    merges = [i for i, r in enumerate(rows) if "nms_merge_kernel" in r["Kernel_Name"]]
    if len(merges) <= warmup:
        raise SystemExit("%s holds %d steps, warm-up is %d" % (path, len(merges), warmup))
    first = merges[warmup - 1] + 1 if warmup > 0 else 0
    steps = len(merges) - warmup
    acc = defaultdict(lambda: [0, 0.0])
    for r in rows[first:merges[-1] + 1]:
        k = short(r["Kernel_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"]) * 1024.0
    return acc, steps


def main():
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    fetch, steps = per_kernel(sys.argv[1], "FETCH_SIZE", warmup)
    write, steps_w = per_kernel(sys.argv[2], "WRITE_SIZE", warmup)
    out = {"steps": steps, "note": "bytes per launch; fetch = 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B), "
                                   "write = WRITE_SIZE; separate rocprofv3 --pmc passes", "kernels": {}}
    for k in sorted(fetch):
        n, fb = fetch[k]
        nw, wb = write.get(k, [0, 0.0])
        out["kernels"][k] = {
            "launches_per_step": n / steps,
            "fetch_bytes_per_launch": 2.0 * fb / n,
            "write_bytes_per_launch": (wb / nw) if nw else None,
            "hbm_bytes_per_launch": 2.0 * fb / n + ((wb / nw) if nw else 0.0),
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
