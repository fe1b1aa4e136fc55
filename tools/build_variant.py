"""Build a variant of libmaple_hip_debug.so (the product library plus include/maple_hip_debug.h: -DMAPLE_DEBUG_ABI is always on)
next to the product library: tools/build_variant.py NAME [extra hipcc flags ...]
-> maple_amd/libmaple_hip_NAME.so (objects under build/NAME/).  Run anything with MAPLE_HIP_LIB=<that path> to use it
(tools that open Device(debug=True): MAPLE_HIP_LIB_DEBUG=<that path> as well -- runtime.load_library keeps the two apart)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, extra = sys.argv[1], ["-DMAPLE_DEBUG_ABI"] + sys.argv[2:]
out = os.path.join(ROOT, "build", name)
os.makedirs(out, exist_ok=True)
objs, jobs = [], []
for unit in g.HIP_UNITS:
    obj = os.path.join(out, unit[:-4] + ".o")
    objs.append(obj)
    jobs.append((unit, subprocess.Popen(["hipcc"] + g.HIP_FLAGS + extra + ["-c", os.path.join(g.CSRC, unit), "-o", obj])))
for unit, job in jobs:
    if job.wait() != 0:
        raise SystemExit(f"hipcc failed on {unit}")
lib = os.path.join(ROOT, "maple_amd", f"libmaple_hip_{name}.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
print(lib)
