#!/usr/bin/env python3
"""Checks the contract between flash_attn_r64m_kernel and the compiler on the kernel's ISA (hipcc -S): outside the
`asm volatile` statements (;;#ASMSTART .. ;;#ASMEND) no instruction of the kernel may name a VGPR above v31 or any AGPR -
those belong to the hand-allocated statements (dvd_amd/csrc/gen_attn_r64m.py) - and the kernel uses no scratch.
usage: check_r64m_isa.py <file.s> [<compiler vgprs> <kernel name>]   (exit 0 = holds; prints the offending lines otherwise)"""
import re
import sys


def check(text, compiler_vgprs=32, kernel="flash_attn_r64m_kernel"):
    bad, kernels = [], 0
    for m in re.finditer(r"^(_ZN3dvd\d+" + kernel + r"\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        kernels += 1
        inside = False
        for ln in m.group(2).split("\n"):
            if "#ASMSTART" in ln:
                inside = True
            elif "#ASMEND" in ln:
                inside = False
            elif not inside:
                code = ln.split(";")[0]
                if "scratch_" in code:
                    bad.append((m.group(1), ln.strip()))
                for a, b in re.findall(r"\b[va]\[(\d+):(\d+)\]", code):
                    if int(b) >= compiler_vgprs or re.search(r"\ba\[", code):
                        bad.append((m.group(1), ln.strip()))
                for r_ in re.findall(r"\bv(\d+)\b", code):
                    if int(r_) >= compiler_vgprs:
                        bad.append((m.group(1), ln.strip()))
                if re.search(r"\ba\d+\b", code) or "accvgpr" in code:
                    bad.append((m.group(1), ln.strip()))
    return kernels, bad


if __name__ == "__main__":
    k, bad = check(open(sys.argv[1]).read(), *([int(sys.argv[2]), sys.argv[3]] if len(sys.argv) > 3 else []))
    for name, ln in bad[:40]:
        print(f"{name}: {ln}")
    print(f"{k} kernel(s) checked, {len(bad)} violation(s)")
    sys.exit(1 if bad or not k else 0)
