"""Register / LDS / scratch use of every kernel in a built library, read from the gfx950 code object's metadata notes.

    python3 tools/regs.py [path/to/lib.so or .o] [substring ...]      default: molvoxel_amd/csrc/mvx_kernels.o
"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "molvoxel_amd/csrc/mvx_kernels.o")
pats = [a for a in sys.argv[1:] if not os.path.exists(a)]
with tempfile.TemporaryDirectory() as td:
    co = os.path.join(td, "k.co")
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "ignored")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows = []
for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
    def g(key):
        m = re.search(rf"\.{key}:\s*(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    rows.append((name, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                 g("group_segment_fixed_size")))
for r in sorted(rows):
    short = re.sub(r"\(.*", "", r[0]).replace("void mvx::", "")
    if pats and not any(p in short for p in pats):
        continue
    print(f"{short:75s} vgpr {r[1]:>4s} sgpr {r[2]:>4s} vspill {r[3]:>3s} sspill {r[4]:>3s} scratch {r[5]:>4s} lds {r[6]:>6s}")
