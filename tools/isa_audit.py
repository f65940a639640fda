#!/usr/bin/env python3
"""Instruction audit of one PGS row visit: disassembles the shipped lean code object
(mujoco_amd/csrc/build/mjh_kern_lean.o), finds the sweep loop of a `solve_pgs_fast<1, L, NT>` instance (the exact,
register-resident sweep with AR in LDS) and of `solve_pgs_resid<1>` (the opt-in residual-update sweep) and classifies
the instructions of one trip through it.  Runs anywhere (no GPU).

  python tools/isa_audit.py [L NT] ...          default: the instances for nefc = 9, 16, 31, 44 (L = nefc / 4, NT = nefc % 4)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble():
    obj = os.path.join(ROOT, "mujoco_amd", "csrc", "build", "mjh_kern_lean.o")
    td = tempfile.mkdtemp()
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "lean.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                    f"--output={co}", "--unbundle"], check=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout.splitlines()


def functions(lines):
    out, cur = {}, None
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur and ln.startswith("\t"):
            out[cur].append(ln)
    return out


def parse(body):
    ins = []
    for ln in body:
        m = re.match(r"\t(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def loops(ins):
    """(start index, end index) of every backward branch"""
    adr = {a: i for i, (a, _, _) in enumerate(ins)}
    out = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            off = int(args.split()[0])
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4*off
            if tgt <= a and tgt in adr:
                out.append((adr[tgt], i))
    return out


CLASSES = [
    ("f64 arithmetic (v_add/mul/fma_f64)", lambda op, a: re.match(r"v_(add|mul|fma|max|min)_f64", op)),
    ("DPP row broadcast (v_mov_b64_dpp)", lambda op, a: "dpp" in op or "dpp" in a or "row_newbcast" in a),
    ("cross-lane to scalar (v_readlane / v_readfirstlane)", lambda op, a: op.startswith(("v_readlane", "v_readfirstlane"))),
    ("scalar -> vector moves (v_mov_b32 from SGPR)", lambda op, a: op.startswith("v_mov_b32") or op.startswith("v_mov_b64")),
    ("compare / select (v_cmp, v_cndmask)", lambda op, a: op.startswith(("v_cmp", "v_cndmask"))),
    ("address / integer VALU", lambda op, a: op.startswith("v_")),
    ("LDS / memory (ds_, global_, flat_, s_load)", lambda op, a: op.startswith(("ds_", "global_", "flat_", "s_load", "buffer_"))),
    ("waits / nops (s_waitcnt, s_nop)", lambda op, a: op.startswith(("s_waitcnt", "s_nop"))),
    ("scalar ALU / branches", lambda op, a: op.startswith("s_")),
]


def audit(name, ins, want):
    best = None
    for a, b in loops(ins):
        n = b - a + 1
        if n > 220:
            continue
        score = sum(1 for _, op, _ in ins[a:b + 1] if want(op))
        if best is None or score > best[0] or (score == best[0] and n < best[1]):
            best = (score, n, a, b)
    if best is None:
        print(f"{name}: no loop found")
        return
    _, n, a, b = best
    counts = [0]*len(CLASSES)
    for _, op, args in ins[a:b + 1]:
        for k, (_, f) in enumerate(CLASSES):
            if f(op, args):
                counts[k] += 1
                break
    valu = sum(counts[:6])
    print(f"{name}: {n} instructions per row visit, {valu} of them vector ALU")
    for (label, _), c in zip(CLASSES, counts):
        print(f"    {c:4d}  {label}")


def main():
    lines = disassemble()
    fn = functions(lines)
    dem = subprocess.run(["c++filt"], input="\n".join(fn), capture_output=True, text=True).stdout.splitlines()
    by_dem = dict(zip(dem, fn))
    args = [int(x) for x in sys.argv[1:]]
    cases = list(zip(args[0::2], args[1::2])) or [(2, 1), (4, 0), (7, 3), (11, 0)]
    print("# one trip through the PGS sweep loop (a row visit) in the shipped gfx950 code object; AR in LDS")
    for L, NT in cases:
        key = [d for d in by_dem if f"solve_pgs_fast<1, {L}, {NT}>" in d]
        if key:
            audit(f"exact sweep, nefc = {4*L + NT} (solve_pgs_fast<1, {L}, {NT}>: chain of {L} dependent adds)", parse(fn[by_dem[key[0]]]),
                  lambda op: op.startswith("v_add_f64"))
    key = [d for d in by_dem if "solve_pgs_resid<1>" in d]
    if key:
        audit("residual-update sweep, any nefc <= 64 (solve_pgs_resid<1>, opt-in)", parse(fn[by_dem[key[0]]]), lambda op: op.startswith(("v_fma_f64", "v_mul_f64")))


if __name__ == "__main__":
    main()
