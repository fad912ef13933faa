"""mvx_plan_call (the pure host decision function) under UBSan over extreme call shapes - CPU only, no GPU:
    bash tools/plan_fuzz_ubsan.sh      (builds molvoxel_amd/csrc/ab/libplan_ubsan.so with -fsanitize=undefined, then runs this)"""
import ctypes as C, sys, random
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt='/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.ubsan_standalone-x86_64.so'
C.CDLL(rt, mode=C.RTLD_GLOBAL)
from molvoxel_amd.voxelizer.hip import _lib
lib=C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'molvoxel_amd/csrc/ab/libplan_ubsan.so'))
lib.mvx_plan_call.restype=C.c_int
random.seed(1)
n=0
for it in range(300000):
    D=random.choice([1,2,3,4,5,7,8,15,16,24,33,48,63,64,65,72,96,100,127,128,129,160,200,255,256,511,512,1000,1024])
    Cc=random.choice([1,2,3,4,5,8,15,16,17,31,32,33,40,63,64,65,128,256,1000,4096])
    B=random.choice([1,2,3,4,8,16,63,64,65,96,128,256,1000,4096,65535,65536,100000,1<<20])
    atoms=random.choice([0,1,8,50,1000,4000,10000,100000,1<<20,1<<30])
    total=min(B*atoms,(1<<62))
    q=_lib.MvxPlanQuery(D, random.choice([0,1,4,5,8,12,16,64,D]), random.choice([32,64]), random.choice([0,1,2]), random.choice([0,1,2]), B, Cc, random.choice([0,1]), total, atoms)
    p=_lib.MvxPlan()
    rc=lib.mvx_plan_call(C.byref(q),C.byref(p))
    n+=1
print("queries", n, "no undefined behaviour reported")
