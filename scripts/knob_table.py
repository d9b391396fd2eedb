import sys
sys.path.insert(0,'/root/repo/dmft-lanc-ed_amd')
from edgpu import capi
ks=capi.knobs()
lines=[]
for kind,title in (("form","Forms the tests force on small sectors (defaults decided by size)"),("fallback","Fallbacks (the default is the measured winner; the other value is exercised by the test named)"),("multi-GPU","Several ranks: forms awaiting a measurement on real xGMI links"),("test hook","Test hooks (geometry of launches, memory caps)"),("debug","Diagnostics")):
    lines.append("* *%s*"%title)
    for k in ks:
        if k["kind"]==kind:
            lines.append("  * `%s` (%s) — %s — `tests/%s`"%(k["name"],"auto" if k["default"]==-1 else k["default"],k["doc"],k["covered_by"]))
new="\n".join(lines)+"\n"
p='/root/repo/scripts/README.md'
s=open(p).read()
a=s.index("* *Forms the tests force"); b=s.index("\nTuning values that were measured and fixed")
s=s[:a]+new+s[b:]
open(p,'w').write(s)
