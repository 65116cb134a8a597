#!/usr/bin/env python3
"""Registers, spills and LDS of the library's kernels, from the code objects of the last build (egotap_amd/build/*.o).
usage: python tools/kernel_regs.py [substring ...]      (no argument: every kernel with a spill or >= 200 VGPRs)"""
import glob, os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
pats = sys.argv[1:]
for obj in sorted(glob.glob(os.path.join(os.environ.get("KREGS_DIR", os.path.join(REPO, "egotap_amd", "build")), "*.o"))):
    with tempfile.TemporaryDirectory() as td:
        import shutil
        tmp_obj = os.path.join(td, "o.o")
        shutil.copy(obj, tmp_obj)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp_obj], check=True, capture_output=True)      # writes <obj>.0.<target> next to it
        co = glob.glob(tmp_obj + ".*gfx950")[0]
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
        vg, sp = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)), int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        ag = int(re.match(r":\s+(\d+)", blk).group(1)) if re.match(r":\s+(\d+)", blk) else 0
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
        if (pats and any(p in dem for p in pats)) or (not pats and (sp > 0 or vg >= 200)):
            print(f"{os.path.basename(obj)[-8:-2]} vgpr {vg:3d} agpr {ag:3d} spill {sp:3d} lds {lds:6d}  {dem[:150]}")
