"""Registers / LDS / scratch of the compiled gfx950 kernels, read from the code-object metadata (no GPU needed):

    python tools/kernel_resources.py tdr_umap_sched.hip [name-filter]

compiles the file for the device only, unbundles the code object and prints one line per kernel."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
src = os.path.join(ROOT, "torchdr_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    obj, dev = os.path.join(tmp, "a.o"), os.path.join(tmp, "dev.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-c", src, "-o", obj,
                    "-I" + os.path.join(ROOT, "include")], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev], check=True)
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", dev], capture_output=True, text=True, check=True).stdout
cxxfilt = shutil.which("c++filt")
for k in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
    name = re.search(r"\.name:\s+(\S+)", k).group(1)
    if cxxfilt:
        name = subprocess.run([cxxfilt, name], capture_output=True, text=True).stdout.strip()
    if flt not in name:
        continue

    def g(key):
        m = re.search(rf"\.{key}:\s+(\d+)", k)
        return int(m.group(1)) if m else 0

    v = g("vgpr_count")
    waves = min(8, 512 // max(v, 1)) if v else 8
    print(f"{name[:110]:110s} vgpr {v:3d} sgpr {g('sgpr_count'):3d} lds {g('group_segment_fixed_size'):6d} scratch "
          f"{g('private_segment_fixed_size'):4d} spills v{g('vgpr_spill_count')}/s{g('sgpr_spill_count')} -> <= {waves} waves/SIMD")
