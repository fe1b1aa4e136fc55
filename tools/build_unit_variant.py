"""A variant of the product library in which only SOME translation units are compiled with extra flags (the others are the objects
of the regular build): tools/build_unit_variant.py NAME unit.hip[,unit2.hip] [extra hipcc flags ...]
-> maple_amd/libmaple_hip_NAME.so.  Run anything with MAPLE_HIP_LIB=<that path> to use it."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, units, extra = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
out = os.path.join(ROOT, "build", name)
os.makedirs(out, exist_ok=True)
objs, jobs = [], []
for unit in g.HIP_UNITS:
    if unit in units:
        obj = os.path.join(out, unit[:-4] + ".o")
        jobs.append((unit, subprocess.Popen(["hipcc"] + g.HIP_FLAGS + extra + ["-c", os.path.join(g.CSRC, unit), "-o", obj])))
    else:
        obj = os.path.join(g.CSRC, unit[:-4] + ".o")
    objs.append(obj)
for unit, job in jobs:
    if job.wait() != 0:
        raise SystemExit(f"hipcc failed on {unit}")
lib = os.path.join(ROOT, "maple_amd", f"libmaple_hip_{name}.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
print(lib)
