#!/usr/bin/env python3
"""Model only (no GPU): LRU model of one XCD L2 for the Ns=18 dw pass of ONE column tile -- the shipped walk (16-row chunks, 256
workgroups in flight, result stream passing through) against a block-tiled walk (rows with equal top-t bits staged in LDS, only the
cross-block partners through L2).  l2_dw_model.py TILE_WIDTH   -- DESIGN.md section 4 quotes the result."""
import numpy as np, itertools, sys
from collections import OrderedDict
Ns=18; n=9
states=np.array(sorted(sum(1<<b for b in c) for c in itertools.combinations(range(Ns),n)),dtype=np.int64)
D=len(states); rank={int(s):i for i,s in enumerate(states)}
hops=[[] for _ in range(D)]
for i,m in enumerate(states):
    m=int(m)
    for b in range(1,Ns):
        if (m&1) and not (m>>b)&1: hops[i].append(rank[m-1+(1<<b)])
        elif not (m&1) and (m>>b)&1: hops[i].append(rank[m+1-(1<<b)])
class LRU:
    def __init__(s,cap): s.cap=cap; s.d=OrderedDict(); s.miss=0; s.acc=0
    def touch(s,k,reuse=True):
        s.acc+=1
        if k in s.d: s.d.move_to_end(k); return
        s.miss+=1; s.d[k]=1
        if len(s.d)>s.cap: s.d.popitem(last=False)
def sim_current(capseg, R=16, inflight=256, pollute=2):
    # one tile; entries = (row) segments; streams: out read (+write) modelled as `pollute` unique entries per row
    L=LRU(capseg); nch=(D+R-1)//R; uid=0
    # interleave: groups of `inflight` chunks proceed row by row (round-robin over chunks)
    for g0 in range(0,nch,inflight):
        chs=list(range(g0,min(nch,g0+inflight)))
        for r in range(R):
            for c in chs:
                i=c*R+r
                if i>=D: continue
                for j in hops[i]: L.touch(('x',j))
                for p in range(pollute): uid+=1; L.touch(('o',uid))
    gather_miss=L.miss-uid
    return gather_miss/D
def sim_block(capseg, t, inflight=32, order="asc"):
    # blocks = rows with equal top-t bits of the 18-bit word (contiguous ranges); own rows staged in LDS (counted 1 HBM read each,
    # passing through L2 once: we insert them), cross-block partners through L2
    shift=Ns-t
    top=(states>>shift)
    bounds=np.flatnonzero(np.diff(top))+1; starts=np.concatenate(([0],bounds)); ends=np.concatenate((bounds,[D]))
    blocks=list(zip(starts,ends))
    if order=="gray":
        pats=[int(top[a]) for a,b in blocks]; key=[p^(p>>1) for p in pats]   # not a true gray walk but groups neighbours
        blocks=[blk for _,blk in sorted(zip(key,blocks))]
    L=LRU(capseg); own=0; maxrows=0
    for g0 in range(0,len(blocks),inflight):
        grp=blocks[g0:g0+inflight]
        maxlen=max(b-a for a,b in grp)
        for r in range(maxlen):
            for a,b in grp:
                i=a+r
                if i>=b: continue
                L.touch(('x',i))      # own row fill
                for j in hops[i]:
                    if not (a<=j<b): L.touch(('x',j))
        maxrows=max(maxrows,maxlen)
    return L.miss/D, maxrows, len(blocks)
def sim_order(capseg, chunk_order, R=16, inflight=256, pollute=2):
    """the shipped walk with the 16-row chunks taken in a given ORDER (round 4: does a Gray-code walk over the top-bit segments help?)"""
    L=LRU(capseg); uid=0
    for g0 in range(0,len(chunk_order),inflight):
        chs=chunk_order[g0:g0+inflight]
        for r in range(R):
            for c in chs:
                i=c*R+r
                if i>=D: continue
                for j in hops[i]: L.touch(('x',j))
                for p in range(pollute): uid+=1; L.touch(('o',uid))
    return (L.miss-uid)/D
def gray_orders(cap, R=16):
    nch=(D+R-1)//R
    print("chunk order natural: gather misses per row %.2f"%sim_order(cap,list(range(nch))))
    for t in (3,4,5,6,7):
        top=(states>>(Ns-t)); pat=np.array([int(top[min(c*R,D-1)]) for c in range(nch)])
        gray=[i^(i>>1) for i in range(1<<t)]; pos={g:i for i,g in enumerate(gray)}
        order=sorted(range(nch), key=lambda c:(pos[pat[c]], c))
        snake=[]
        for k,g in enumerate(gray):
            cs=[c for c in range(nch) if pat[c]==g]; snake+= cs if k%2==0 else cs[::-1]
        print("chunk order gray over top-%d bits: %.2f   as a snake: %.2f"%(t,sim_order(cap,order),sim_order(cap,snake)))
if __name__=="__main__":
    w=int(sys.argv[1]); segB=w*8; cap=4*1024*1024//segB
    print("tile width",w,"L2 capacity in row segments",cap)
    for R,infl in ((16,256),(16,128)):
        print("current R=%d inflight=%d: gather misses per row %.2f (+1 out read) -> read B/elt %.1f"%(R,infl,sim_current(cap,R,infl),0))
    for t in (6,7,8):
        for infl in (32,16):
            m,mr,nb=sim_block(cap,t,infl)
            print("block t=%d inflight=%d: misses per row %.2f  max rows/block %d (LDS %.0f KB) blocks %d"%(t,infl,m,mr,mr*segB/1024,nb))
    gray_orders(cap)
