import subprocess,sys,re
"""Registers, spills, scratch and LDS of every kernel of a code object: tools/kernel_meta.py FILE [name filter].
FILE: a gfx950 ELF, an offload bundle (hipcc --cuda-device-only -c) or a host object with a .hip_fatbin section."""
import os,tempfile
BIN="/opt/rocm/lib/llvm/bin/"
src=sys.argv[1]
tmp=tempfile.mkdtemp()
if open(src,"rb").read(4)==b"\x7fELF" and b".hip_fatbin" in subprocess.check_output([BIN+"llvm-readelf","-S",src]):
    subprocess.check_call([BIN+"llvm-objcopy","--dump-section",".hip_fatbin="+tmp+"/fat",src]); src=tmp+"/fat"
if open(src,"rb").read(4)!=b"\x7fELF":
    subprocess.check_call([BIN+"clang-offload-bundler","--unbundle","--type=o","--input="+src,"--targets=hipv4-amdgcn-amd-amdhsa--gfx950","--output="+tmp+"/co"]); src=tmp+"/co"
out=subprocess.check_output([BIN+"llvm-readelf","--notes",src],text=True)
ks=[];cur={}
for line in out.splitlines():
    m=re.match(r"\s+(-\s+)?\.(\w+):\s+(.*)",line)
    if not m: continue
    k,v=m.group(2),m.group(3)
    if k=="agpr_count" or (m.group(1) and cur.get("name")):
        pass
    if m.group(1) and "name" in cur and "vgpr_count" in cur:
        ks.append(cur);cur={}
    if k in("name","vgpr_count","agpr_count","sgpr_count","private_segment_fixed_size","vgpr_spill_count","sgpr_spill_count","group_segment_fixed_size","max_flat_workgroup_size"):
        cur[k]=v
if "vgpr_count" in cur: ks.append(cur)
flt=sys.argv[2] if len(sys.argv)>2 else ""
for k in ks:
    n=subprocess.check_output(["c++filt",k["name"]],text=True).strip()
    n=re.sub(r"\(anonymous namespace\)::","",n); n=n.split("(")[0]
    if flt in n:
        print(f'{n[:70]:70s} vgpr {k["vgpr_count"]:>4s} agpr {k.get("agpr_count","-"):>4s} sgpr {k["sgpr_count"]:>4s} scratch {k["private_segment_fixed_size"]:>5s} vspill {k["vgpr_spill_count"]:>4s} sspill {k["sgpr_spill_count"]:>4s} lds {k["group_segment_fixed_size"]:>6s}')
