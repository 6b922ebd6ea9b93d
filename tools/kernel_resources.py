#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in the built library (from the gfx950 code objects' metadata notes)."""
import os, re, subprocess, sys, tempfile
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "radae_amd", "libradehip.so")
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
    data = open(fat, "rb").read()
    # the section holds one offload bundle per translation unit: split at the bundle magic
    offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    for k, o in enumerate(offs):
        part = os.path.join(td, f"b{k}.bin"); open(part, "wb").write(data[o:offs[k + 1] if k + 1 < len(offs) else len(data)])
        co = os.path.join(td, f"d{k}.co")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
        if r.returncode or not os.path.exists(co): continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count: (\d+)", notes, re.S):
            print(f"{m.group(2)[:70]:72s} vgpr {m.group(5):>3s} spill {m.group(6):>3s} scratch {m.group(3):>4s} B static-lds {m.group(1):>6s}")
