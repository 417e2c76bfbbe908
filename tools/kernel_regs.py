#!/usr/bin/env python3
"""Register / scratch / LDS use per kernel of a built object: tools/kernel_regs.py raytracer_amd/csrc/rt_kernels.o [filter]"""
import os, re, subprocess, sys, tempfile, glob
obj = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
LL = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as td:
    tmp = os.path.join(td, "o.o"); subprocess.run(["cp", obj, tmp], check=True)
    subprocess.run([LL + "llvm-objdump", "--offloading", tmp], check=True, capture_output=True, cwd=td)
    co = [f for f in glob.glob(tmp + ".*") if "amdgcn" in f][0]
    notes = subprocess.run([LL + "llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
for b in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if flt and flt not in name: continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    dem = subprocess.run([LL + "llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() if os.path.exists(LL + "llvm-cxxfilt") else name
    print("%-90s vgpr %3s agpr %s sgpr %3s spill %3s scratch %5s lds %6s" % (dem[:90], g("vgpr_count"), b.split(":")[1].split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
