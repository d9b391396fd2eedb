#!/usr/bin/env python3
"""Model only (no GPU): the per-pair row matrix of the needed-rows exchange for the half-filled Ns-site sector on P ranks, and what a
two-phase store-and-forward relay of the heavy pairs through idle peers would do to the busiest link (rows; one hop-1 and one hop-2
message per directed link; direct traffic split over both phases).  relay_model.py NS P   -- DESIGN.md section 6 quotes the result."""
import numpy as np, itertools, sys
def matrix(Ns, n, P):
    states=np.array(sorted(sum(1<<b for b in c) for c in itertools.combinations(range(Ns),n)),dtype=np.int64)
    D=len(states); rank={int(s):i for i,s in enumerate(states)}
    q,rem=divmod(D,P)
    owner=np.zeros(D,int); d0=[]
    for r in range(P):
        a=r*(q+1) if r<rem else r*q+rem; b=a+(q+1 if r<rem else q); owner[a:b]=r; d0.append((a,b))
    need=[set() for _ in range(P)]
    for i,m in enumerate(states):
        m=int(m); r=owner[i]
        for b in range(1,Ns):
            if (m&1) and not (m>>b)&1: j=rank[m-1+(1<<b)]
            elif not (m&1) and (m>>b)&1: j=rank[m+1-(1<<b)]
            else: continue
            if owner[j]!=r: need[r].add(j)
    M=np.zeros((P,P),int)   # M[s][d]: rows s sends to d
    for d in range(P):
        for j in need[d]: M[owner[j]][d]+=1
    return M,D
def route(M):
    P=M.shape[0]
    dirn=M.astype(float).copy()
    L1=dirn/2; L2=dirn/2
    used1=np.zeros((P,P),bool); used2=np.zeros((P,P),bool)
    relays=[]
    for it in range(4*P*P):
        T=L1.max()+L2.max()
        # pair with the largest direct load
        s,d=np.unravel_index(np.argmax(np.maximum(L1,L2)*(dirn>0)),L1.shape)
        Ld=max(L1[s,d],L2[s,d])
        best=None
        for r in range(P):
            if r in (s,d) or used1[s,r] or used2[r,d]: continue
            base=max(L1[s,r],L2[r,d])
            delta=(Ld-base)/1.5
            if delta<1: continue
            delta=min(delta,dirn[s,d])
            if best is None or delta>best[1]: best=(r,delta)
        if best is None: break
        r,delta=best
        L1[s,d]-=delta/2; L2[s,d]-=delta/2; dirn[s,d]-=delta
        L1[s,r]+=delta; L2[r,d]+=delta; used1[s,r]=True; used2[r,d]=True
        relays.append((s,d,r,delta))
        if L1.max()+L2.max()>=T-1e-9 and it>P*P: break
    return L1,L2,relays
if __name__=="__main__":
    Ns=int(sys.argv[1]); P=int(sys.argv[2])
    M,D=matrix(Ns,Ns//2,P)
    print("rows per rank",D//P,"received per rank",M.sum(0),"busiest pair",M.max(),"balanced per link",M.sum(0).max()/(P-1))
    print(M)
    L1,L2,rel=route(M)
    print("T before",M.max(),"after",L1.max()+L2.max(),"relays",len(rel))
    print(np.round(L1+L2).astype(int))
