"""Scans the gfx950 code of a built library for an instruction form that gfx950 executes wrongly.

Measured with tools/probe_pkfma.hip and tools/probe_pkfma2.hip on MI355X: a packed fp32 FMA
    v_pk_fma_f32 D, s[n:n+1], V1, V2  op_sel:[0,0,1] op_sel_hi:[0,...]
whose multiplier (first or second source) is a scalar register pair read as (low, low) and whose
LOW result takes its addend from the HIGH register of the addend pair (op_sel bit of the third
source set) returns that low result WITHOUT the addend on the last 16-lane passes of the wave
(lanes 32-63) when, at that moment, the matrix pipe of the same SIMD is executing another wave's
MFMAs.  Alone, or beside vector-only kernels, the instruction is exact; with the scalar pair read
as a real pair (op_sel_hi set), with a vector or constant multiplier, with the swap on another
source, as v_pk_mul_f32 / v_pk_add_f32, or with an `s_nop 7` in front of it, it is exact too.
hipcc forms the instruction itself (SLP vectorisation of scalar code + folding of a scalar
operand), so no source-level idiom rules it out; this check does, on the machine code.

A finding = any v_pk_fma_f32 with a scalar-register source and the addend's low-half selector
set (a superset of the measured failing forms).

    python tools/isa_lint.py [library.so ...]        exit status 1 if anything is found
"""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
_SRC = re.compile(r"\b(v_pk_fma_f32)\s+(.*)")


def _fields(rest):
    """operands and modifier lists of one disassembled VOP3P line"""
    mods = dict((m.group(1), [int(b) for b in m.group(2).split(",")])
                for m in re.finditer(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[([01,]+)\]", rest))
    ops = re.sub(r"\s*(op_sel_hi|op_sel|neg_lo|neg_hi):\[[01,]+\]", "", rest.split("//")[0]).strip()
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", ops)]
    return ops, mods


def hazardous(line):
    m = _SRC.search(line)
    if not m:
        return False
    ops, mods = _fields(m.group(2))
    if len(ops) != 4:
        return False
    op_sel = mods.get("op_sel", [0, 0, 0])
    scalar = any(re.match(r"(s\[|s\d|vcc|ttmp|exec)", o) for o in ops[1:3])
    return scalar and len(op_sel) == 3 and op_sel[2] == 1


def disassemble(lib):
    """text of every gfx950 code object bundled in `lib`"""
    work = tempfile.mkdtemp(prefix="isa_lint_")
    try:
        copy = os.path.join(work, os.path.basename(lib))
        shutil.copy(lib, copy)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", copy], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = []
        for f in sorted(os.listdir(work)):
            if "amdgcn" in f:
                out.append(subprocess.run([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", os.path.join(work, f)],
                                          check=True, capture_output=True, text=True).stdout)
        return "\n".join(out)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def scan(lib):
    """[(kernel symbol, instruction text)] of every finding in `lib`; also the number of kernels seen"""
    found, kernels, sym = [], 0, "?"
    for line in disassemble(lib).splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            sym = m.group(1)
            kernels += 1
            continue
        if "v_pk_fma_f32" in line and hazardous(line):
            found.append((sym, line.split("//")[0].strip()))
    return found, kernels


if __name__ == "__main__":
    libs = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo_amd", "libymx.so")]
    bad = 0
    for lib in libs:
        found, kernels = scan(lib)
        print("%s: %d functions, %d findings" % (lib, kernels, len(found)))
        for sym, ins in found:
            print("  %s\n      %s" % (sym, ins))
        bad += len(found)
    sys.exit(1 if bad else 0)
