#!/usr/bin/env python3
"""Build-time check of the hand-counted prefetch of the pipelined kernels (pipe_kernel.hip, pipe16_kernel.hip).

The segment prefetch is issued from inline asm and waited for by hand-counted `s_waitcnt vmcnt(k)` statements that
name the loaded register.  The compiler takes an asm output for available at once, so nothing stops it from COPYING
such a register while its load is still in flight (it did, in the order-2 instantiation: phi copies of the second
prefetch buffer; the copies held whatever the registers held before).  This script reads the ISA of every
instantiation and fails unless every tile write that follows a hand-counted wait stores a register that a
nontemporal load of the same kernel writes, i.e. unless the data goes from the load to the tile write without a copy.

usage: check_prefetch_regs.py file.s [file.s ...]      (hipcc -S --cuda-device-only output)"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return tuple(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return (int(m.group(1)),) if m else ()


def check(path):
    bad = 0
    name, lines = None, []
    kernels = []
    for line in open(path):
        m = re.match(r"^(_ZN3dsp\d+iir_pipe\w*kernel\w+):", line)
        if m:
            name, lines = m.group(1), []
            continue
        if name is not None:
            lines.append(line.strip())
            if "s_endpgm" in line:
                kernels.append((name, lines))
                name = None
    for name, lines in kernels:
        loaded = set()
        for l in lines:
            m = re.match(r"global_load_dwordx4 (v\[\d+:\d+\]), .* nt$", l)
            if m:
                loaded.add(regs(m.group(1)))
        if not loaded:
            continue                                  # an instantiation without the hand-counted path
        n_ok = n_bad = 0
        for i, l in enumerate(lines):
            if not re.match(r"s_waitcnt vmcnt\(\d+\)$", l) or i == 0 or not lines[i - 1].startswith(";;#ASMSTART"):
                continue                              # the compiler's own waits look after themselves
            for nxt in lines[i + 1:i + 4]:
                m = re.match(r"ds_write_b128 v\d+, (v\[\d+:\d+\])", nxt)
                if m:
                    if regs(m.group(1)) in loaded:
                        n_ok += 1
                    else:
                        n_bad += 1
                    break
        if n_ok + n_bad == 0:
            continue
        status = "ok" if n_bad == 0 else "COPIED PREFETCH REGISTERS"
        if n_bad:
            bad += 1
        print("%-20s %s: %d parks from load registers, %d from copies" % (status, name, n_ok, n_bad))
    return bad, len([1 for name, lines in kernels if any(l.startswith(";;#ASMSTART") for l in lines)])


if __name__ == "__main__":
    results = [check(p) for p in sys.argv[1:]]
    if sum(r[1] for r in results) == 0:
        print("no kernel with hand-counted waits found in", sys.argv[1:])
        sys.exit(2)
    sys.exit(1 if sum(r[0] for r in results) else 0)
