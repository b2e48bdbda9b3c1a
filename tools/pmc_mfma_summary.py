"""MFMA pipe utilisation per kernel from a rocprofv3 --pmc pass over bench.py.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out -- python3 bench.py ...
    python tools/pmc_mfma_summary.py out/<host>/<pid>_counter_collection.csv [warmup_steps] > profiles/rNN_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over every SIMD, the cycles its matrix pipe was busy (16 per
mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs.  Utilisation =
busy / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): the fraction of the matrix pipes' cycles, at
the clock the chip actually held, that carried an MFMA.  The shader clock follows from
GRBM_GUI_ACTIVE / 8 / duration.
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


def main():
    warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = list(csv.DictReader(open(sys.argv[1])))
    disp = {}
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]),
                                                   "t1": int(r["End_Timestamp"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    order = sorted(disp.values(), key=lambda d: d["t0"])
    merges = [i for i, d in enumerate(order) if "nms_merge_kernel" in d["name"]]
    first = merges[warmup - 1] + 1 if warmup > 0 else 0
    acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for d in order[first:merges[-1] + 1]:
        a = acc[short(d["name"])]
        a[0] += 1
        a[1] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += d.get("GRBM_GUI_ACTIVE", 0.0)
        a[3] += (d["t1"] - d["t0"]) * 1e-9
    out = {"steps": len(merges) - warmup, "kernels": {}}
    for k in sorted(acc):
        n, busy, gui, sec = acc[k]
        if busy <= 0 or gui <= 0:
            continue
        out["kernels"][k] = {"launches_per_step": n / out["steps"], "mfma_utilisation": busy / (gui / 8.0 * 1024.0),
                             "shader_clock_ghz": gui / 8.0 / sec / 1e9, "ms_per_step_profiled": sec * 1e3 / out["steps"]}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
