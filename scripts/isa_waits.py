"""Static look at a kernel's ISA: for every vector-memory load, how many instructions pass before the next `s_waitcnt vmcnt`
and which count it waits for.  A load followed within a few instructions by a low count is a prefetch the compiler defeated
(typically: the value is merged with an older one under a branch and moved to other registers at once).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/edgpu.s dmft-lanc-ed_amd/csrc/edgpu.hip
       python scripts/isa_waits.py /tmp/edgpu.s k_uppass_halo [substring of the mangled template arguments]"""
import re, sys

src = open(sys.argv[1]).read()
pat = sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else ""
for m in re.finditer(r"^(_ZN5edgpu\w*%s\w*):" % re.escape(pat), src, re.M):
    name = m.group(1)
    if sub not in name:
        continue
    body = src[m.end():src.index(".Lfunc_end", m.end())]
    ins = [l.strip() for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((";", "."))]
    print(name, len(ins), "instructions")
    cnt = {}
    for i, l in enumerate(ins):
        op = l.split()[0]
        cnt[op] = cnt.get(op, 0) + 1
        if op.startswith(("global_load", "buffer_load", "flat_load")):
            for j in range(i + 1, min(i + 400, len(ins))):
                w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins[j])
                if w:
                    between = sum(1 for x in ins[i + 1:j] if x.split()[0].startswith(("global_load", "global_store")))
                    print("  %5d %-22s -> wait vmcnt(%s) after %3d instructions (%d memory ops between)" % (i, op, w.group(1), j - i, between))
                    break
    print("  ", {k: v for k, v in cnt.items() if k.startswith(("global_", "ds_", "s_barrier", "s_waitcnt", "v_mov", "scratch"))})
