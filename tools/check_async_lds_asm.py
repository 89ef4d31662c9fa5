#!/usr/bin/env python3
"""Static check for kernels that issue LDS reads from inline asm (dig_amd/csrc/conv3x3_ws.hip).

An inline-asm `ds_read_b128` returns its data asynchronously, but the compiler believes the destination registers are
defined as soon as the asm statement has executed.  Until the matching inline-asm `s_waitcnt lgkmcnt(N)` nothing may
read or overwrite those registers (a compiler-inserted copy or spill in that window would move stale data, and the
late-arriving LDS data would land in a register that may by then hold something else).  This script walks the
generated ISA and reports every instruction that touches a pending destination.

usage: check_async_lds_asm.py <file.s> [kernel-name-substring]
"""
import re
import sys


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def main():
    text = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    in_kernel, in_asm, pending, bad, nreads = False, False, [], 0, 0
    for ln, line in enumerate(text, 1):
        s = line.strip()
        lab = re.match(r"^([A-Za-z_][\w$.]*):", s)
        if lab and not s.startswith(".L"):
            in_kernel = want in lab.group(1)
            pending = []
        if not in_kernel or not s or s.startswith(";") and "ASM" not in s or s.startswith("."):
            if ";;#ASMSTART" in s:
                in_asm = True
            elif ";;#ASMEND" in s:
                in_asm = False
            continue
        if ";;#ASMSTART" in s:
            in_asm = True
            continue
        if ";;#ASMEND" in s:
            in_asm = False
            continue
        code = s.split(";")[0]
        if in_asm and code.startswith("ds_read_b128"):
            dst = regs(code.split(",")[0])
            hit = [p for p in pending if p & dst]
            if hit:
                print(f"{ln}: asm read overwrites a pending destination: {code}")
                bad += 1
            pending.append(dst)
            nreads += 1
            continue
        if in_asm and code.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n else []
            continue
        m = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", code)
        if m:  # a compiler wait also retires (conservatively correct to model)
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n < len(pending) else pending
            if n == 0:
                pending = []
            continue
        touched = regs(code)
        for p in pending:
            if p & touched:
                print(f"{ln}: touches v{sorted(p & touched)} while an asm LDS read into it is in flight: {code}")
                bad += 1
                break
    print(f"{nreads} inline-asm LDS reads checked, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
