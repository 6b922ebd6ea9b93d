#!/usr/bin/env python3
"""Guard against a register-allocator placement bug seen with ROCm 7.2's clang on the big receiver kernel.

At a control-flow join the compiler re-enables the lanes that skipped the branch with `s_or_b64 exec, exec, sN`.
When a VGPR spill (`scratch_store`) or reload (`scratch_load`) is placed in that join block *ahead of* the
`s_or_b64`, it runs with the lanes of the skipped branch still disabled -- with EXEC = 0 when the whole wavefront
branched around (`s_cbranch_execz`).  The slot then keeps stale data and the later full-EXEC reload hands garbage to
every lane.  In k_rx_sync this silently corrupted a long-lived f64 polynomial coefficient of sincos() (frequency
estimates off by 0.01 Hz) in builds that differed from the good one only by an unrelated reduction helper.

This script disassembles the gfx950 code objects embedded in a built object / shared library and reports every basic
block in which a spill STORE precedes the EXEC restore (`s_or_b64 exec, exec, ...` or `s_or_saveexec_b64`), and every RELOAD there whose
register is still read after the restore (a reload whose value is consumed inside the masked block and overwritten before any later
read -- the compiler re-materialising a short-lived temporary under the branch's own mask -- is correct and is not reported).
Exit status 1 if any is found.

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
    """-> disassembly text of every gfx950 code object (one per translation unit) inside a host object / shared library built by hipcc."""
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            raise RuntimeError(f"{path}: no .hip_fatbin section")
        data = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
        for k, o in enumerate(offs):
            part, co = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"d{k}.co")
            open(part, "wb").write(data[o:offs[k + 1] if k + 1 < len(offs) else len(data)])
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", f"--targets={TARGET}", f"--output={co}"], capture_output=True)
            if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co):
                out.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout)
    if not out:
        raise RuntimeError(f"{path}: no gfx950 code object found")
    return "\n".join(out)


VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(text):
    regs = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def _read_after(ins, k, regs, limit=96):
    """is any register of `regs` read by the instructions right after index k before it is overwritten?  A linear scan of the next
    `limit` instructions, ended by a barrier or an unconditional branch: far-away reads belong to other paths, where the allocator reloads again."""
    live = set(regs); fn = ins[k][3]
    for addr, op, args, f, tgt in ins[k + 1:k + 1 + limit]:
        if f != fn or not live or op in ("s_barrier", "s_branch", "s_endpgm", "s_setpc_b64"):
            break
        ops = [a.strip() for a in args.split(",")] if args else []
        store_like = op.startswith(("scratch_store", "global_store", "ds_write", "buffer_store", "flat_store", "s_", "v_cmp", "v_readlane", "v_readfirstlane", "global_atomic", "ds_max", "ds_add", "ds_xor"))
        srcs = _vregs(",".join(ops if store_like else ops[1:]))
        if srcs & live:
            return True
        if not store_like and ops:
            live -= _vregs(ops[0])
    return False


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
    for k, (addr, op, args, fn, tgt) in enumerate(ins):
        if addr in leaders:
            pend = []
        if op.startswith("scratch_store"):
            pend.append((f"{addr:#x}: {op} {args}", None))
        elif op.startswith("scratch_load"):
            pend.append((f"{addr:#x}: {op} {args}", _vregs(args.split(",")[0])))
        elif (op == "s_or_b64" and args.replace(" ", "").startswith("exec,exec,")) or op == "s_or_saveexec_b64":
            bad = [t for t, regs in pend if regs is None or _read_after(ins, k, regs)]
            if bad:                   # EXEC widens here (join after an if, or the else side taking over the remaining lanes)
                hits.append((fn, addr, bad))
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
