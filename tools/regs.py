"""Register / LDS / scratch use of every kernel in a built object or library, read from the gfx950 code object's metadata.

    python3 tools/regs.py [path/to/lib.so or .o] [substring ...]      default: every kernel object of molvoxel_amd/csrc
"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "molvoxel_amd/csrc")
KERNEL_OBJECTS = [os.path.join(CSRC, f"mvx_{name}.o") for name in ("prep", "slab", "pair", "f64", "splat")
                  if os.path.exists(os.path.join(CSRC, f"mvx_{name}.hip"))]


def kernel_resources(obj=None):
    """{demangled kernel name (without 'void mvx::' and the argument list): dict(vgpr, sgpr, vspill, sspill, scratch, lds)}
    obj = None: every kernel object of the library (one per kernel family)."""
    if obj is None:
        out = {}
        for o in KERNEL_OBJECTS:
            if os.path.exists(o):
                out.update(kernel_resources(o))
        return out
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "k.co"), os.path.join(td, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "ignored")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    blocks = re.split(r"\n\s*- \.agpr_count:", notes)[1:]
    names = [re.search(r"\.name:\s*(\S+)", b).group(1) for b in blocks]
    demangled = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for blk, name in zip(blocks, demangled):
        def g(key):
            m = re.search(rf"\.{key}:\s*(\d+)", blk)
            return int(m.group(1)) if m else -1
        short = re.sub(r"\(.*", "", name).replace("void mvx::", "").replace("mvx::", "")
        out[short] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                          scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    return out


if __name__ == "__main__":
    obj = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else None
    pats = [a for a in sys.argv[1:] if not os.path.exists(a)]
    for short, r in sorted(kernel_resources(obj).items()):
        if pats and not any(p in short for p in pats):
            continue
        print(f"{short:75s} vgpr {r['vgpr']:4d} sgpr {r['sgpr']:4d} vspill {r['vspill']:3d} sspill {r['sspill']:3d} scratch {r['scratch']:4d} lds {r['lds']:6d}")
