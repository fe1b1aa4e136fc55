import subprocess,sys,re
out=subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf","--notes",sys.argv[1]],text=True)
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
