"""Per-step kernel times from a rocprofv3 kernel-trace database (tuning aid):
    python tools/trace_summary.py gpurun_out/prof_train/tr_results.db [steps] [name filter]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    flt = sys.argv[3] if len(sys.argv) > 3 else None
    if flt:
        rows = list(db.execute("select name, grid_x, grid_z, lds_size, (end - start) from kernels where name like ? order by start", ("%" + flt + "%",)))
        n = int(len(rows) / steps)
        for r in rows[-n:]:
            print("%-48s wgs=%-6d z=%-4d lds=%-7d %9.1f us" % (r[0].replace("ymx::", "").replace("void ", "")[:48], r[1] // 256, r[2], r[3], r[4] / 1e3))
        return
    total = 0.0
    for name, calls, dur in db.execute("select name, total_calls, total_duration from top_kernels"):
        total += dur
        if dur / steps > 50:
            print("%-72s %6.1f  %7.2f ms/step" % (name.replace("ymx::", "").replace("void ", "")[:72], calls / steps, dur / steps / 1e3))
    print("total %.2f ms/step" % (total / steps / 1e3))


if __name__ == "__main__":
    main()
