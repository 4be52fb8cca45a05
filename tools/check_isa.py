#!/usr/bin/env python3
"""Static check of the gfx950 code that ships in libdesco_hip.so (run by the Makefile after linking; `make` fails on a hit).

Rule PK-OPSEL.  No packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) may take its LOW result
lane from the HIGH dword of src1 or src2 (VOP3P OP_SEL bit 1 or 2: `op_sel:[x,1]`, `op_sel:[x,1,y]`, `op_sel:[x,y,1]`).
On MI355X that operand selection returns wrong values in lanes 48-63 of the low lane while MFMAs execute on the SIMD:
sporadic, different in every launch, not cured by wait states or by draining the wave's own MFMAs
(tools/micro/pk_f32_probe3.hip reproduces it in isolation; profiles/r5_a_gossip_f16_hazard.md has the bisection: 40
builds of the gossip kernel, every one with such an instruction fails, every one without is bit-reproducible).  The
same arithmetic with the selector on src0 (`op_sel:[1,0]`), with OP_SEL_HI on any source, or without selectors is exact.
hipcc picks the operand order of a commutative packed instruction itself, so the rule cannot be kept by construction in
the source: it is checked on the binary.  A hit is fixed in the source by moving the broadcast scalar out of the odd
register of a 64-bit pair (or by writing that product in scalar form).

usage: check_isa.py <libdesco_hip.so | file.s | file.o> ...        exit code 1 on a hit"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
OPSEL = re.compile(r"op_sel:\[([01])(?:,([01]))?(?:,([01]))?\]")


def bad_line(line):
    if not PK.search(line):
        return False
    m = OPSEL.search(line)
    return bool(m) and (m.group(2) == "1" or m.group(3) == "1")


def scan_text(name, text, hits):
    func = "?"
    n = 0
    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"^(?:[0-9a-f]+ <)?([A-Za-z_][\w.$]*)>?:", s)
        if m and not s.startswith("."):
            func = m.group(1)
        if PK.search(s):
            n += 1
            if bad_line(s):
                hits.append((name, func, s.split("//")[0].strip()))
    return n


def disassemble(path, workdir):
    """device disassembly of every gfx950 code object bundled in a host ELF (.so / .o)"""
    local = os.path.join(workdir, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = []
    for f in sorted(os.listdir(workdir)):
        if "amdgcn" in f and f.startswith(os.path.basename(path) + "."):
            r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(workdir, f)], check=True,
                               capture_output=True, text=True)
            out.append((f, r.stdout))
    if not out:
        raise RuntimeError(f"{path}: no gfx950 code object found")
    return out


def main():
    hits, total = [], 0
    for p in sys.argv[1:]:
        if p.endswith(".s"):
            total += scan_text(p, open(p).read(), hits)
            continue
        with tempfile.TemporaryDirectory() as wd:
            for name, text in disassemble(p, wd):
                total += scan_text(name, text, hits)
    print(f"check_isa: {total} packed-fp32 instructions scanned, {len(hits)} with OP_SEL on src1/src2 (rule PK-OPSEL)")
    for name, func, line in hits[:40]:
        print(f"  {name}: {func}: {line}")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
