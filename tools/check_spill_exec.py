#!/usr/bin/env python3
"""Guard against a register-allocator placement bug seen with ROCm 7.2's clang on the big receiver kernel.

At a control-flow join the compiler re-enables the lanes that skipped the branch with `s_or_b64 exec, exec, sN`.
When a VGPR spill (`scratch_store`) or reload (`scratch_load`) is placed in that join block *ahead of* the
`s_or_b64`, it runs with the lanes of the skipped branch still disabled -- with EXEC = 0 when the whole wavefront
branched around (`s_cbranch_execz`).  The slot then keeps stale data and the later full-EXEC reload hands garbage to
every lane.  In k_rx_sync this silently corrupted a long-lived f64 polynomial coefficient of sincos() (frequency
estimates off by 0.01 Hz) in builds that differed from the good one only by an unrelated reduction helper.

This script disassembles the gfx950 code object embedded in a built object / shared library and reports every basic
block in which a scratch access precedes the EXEC restore (`s_or_b64 exec, exec, ...` or `s_or_saveexec_b64`).  Exit status 1 if any is found.

    python tools/check_spill_exec.py radae_amd/libradehip.so
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def disassemble(path):
    """-> disassembly text of the gfx950 code object inside a host object / shared library built by hipcc."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            raise RuntimeError(f"{path}: no .hip_fatbin section")
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"], check=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout


SYM = re.compile(r"^([0-9a-f]+) <([^>]+)>:")
INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):")
TGT = re.compile(r"<([^>+]+)\+0x([0-9a-f]+)>\s*$")


def scan(text):
    """-> list of (function, address of the s_or_b64, [scratch instructions ahead of it])"""
    base, func, ins = {}, None, []
    for line in text.split("\n"):
        m = SYM.match(line)
        if m:
            func = m.group(2); base[func] = int(m.group(1), 16)
            continue
        m = INS.match(line)
        if m and func:
            t = TGT.search(line)
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), func, (t.group(1), int(t.group(2), 16)) if t else None))
    leaders = set(base.values())
    for k, (addr, op, args, fn, tgt) in enumerate(ins):
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            if tgt and tgt[0] in base:
                leaders.add(base[tgt[0]] + tgt[1])
            if k + 1 < len(ins):
                leaders.add(ins[k + 1][0])
    hits, pend = [], []
    for addr, op, args, fn, tgt in ins:
        if addr in leaders:
            pend = []
        if op.startswith("scratch_store") or op.startswith("scratch_load"):
            pend.append(f"{addr:#x}: {op} {args}")
        elif (op == "s_or_b64" and args.replace(" ", "").startswith("exec,exec,")) or op == "s_or_saveexec_b64":
            if pend:                  # EXEC widens here (join after an if, or the else side taking over the remaining lanes)
                hits.append((fn, addr, pend))
            pend = []
        elif "exec" in args.split(",")[0] or op == "s_barrier" or "saveexec" in op:
            pend = []          # any other EXEC write / barrier: what follows is not the join prologue
    return hits


def main():
    paths = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "radae_amd", "libradehip.so")]
    bad = 0
    for p in paths:
        hits = scan(disassemble(p))
        for fn, addr, pend in hits:
            print(f"{p}: {fn}: EXEC restore at {addr:#x} comes after {len(pend)} scratch access(es), first {pend[0]}")
        print(f"{p}: {len(hits)} join block(s) with a spill ahead of the EXEC restore")
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
