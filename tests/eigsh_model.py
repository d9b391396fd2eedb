"""Numpy model of edgpu_eigsh's thick-restart Lanczos with partial re-orthogonalisation (csrc/lanczos.hpp: eigsh_pro) --
TEST INFRASTRUCTURE.  Same steps, same bookkeeping, vectors as numpy arrays and the product as a callable, so that the
algorithm (omega recurrence for an arrow + tridiagonal projected matrix, selective passes against the kept Ritz vectors,
the relation-defect term eta carried across restarts) can be exercised on the CPU against dense diagonalisation, and
instrumented: `reads` counts basis-vector reads, `verbose` prints the true loss of orthogonality next to its estimate.
`defect=False` switches the eta term off (the first GPU form of the routine, with blind estimates after a few restarts)."""
import numpy as np


def eigsh_pro(matvec, n, nev, ncv, maxiter=300, tol=1e-12, tau=1.5e-8, seed=0, verbose=False, defect=True):
    eps=2.220446049250313e-16; eps23=eps**(2/3); tol_eff=max(tol,1e-12)
    m=ncv; M=m+2
    Bv=[None]*(m+1); sc=np.ones(M); T=np.zeros((M,M)); W=np.zeros((M,M)); F=np.zeros((M,M)); eta=np.zeros(M)
    rng=np.random.RandomState(seed); u0=rng.uniform(0,1,n)
    Bv[0]=u0.copy(); sc[0]=1/np.linalg.norm(u0); W[0,0]=1
    l=0;j=0;normT=0;nmv=0;reads=0;nev_full=0;nev_sel=0
    uprev=np.zeros(n); s_prev=0.0; beta_prev=0.0
    for it in range(maxiter):
        meff=m
        while j<m:
            u=Bv[j]; s_cur=sc[j]
            w=s_cur*matvec(u)-beta_prev*s_prev*uprev; nmv+=1
            alpha=(s_cur*u)@w
            w=w-alpha*s_cur*u
            beta=np.linalg.norm(w)
            T[j,j]=alpha
            normT=max(normT,abs(alpha)+beta+(abs(T[j,j-1]) if j>0 else 0))
            arrow = l>0 and j==l
            idx=[]
            if arrow and beta>0:      # known coefficients sigma_i = T[i,l]: one pass, no dots
                for i in range(l): w=w-T[i,l]*sc[i]*Bv[i]
                reads+=l; beta=np.linalg.norm(w)
            if True:
                psi=2*eps*normT
                for k in range(j):
                    num=0.0
                    for i in range(j+1):
                        if T[i,k]!=0: num+=T[i,k]*W[i,j]
                        if T[i,j]!=0: num-=T[i,j]*W[i,k]
                    src=psi+(eta[k]+eta[j] if defect else 0.0)      # relation defect of column k / j
                    ww=(num+np.copysign(src,num))/beta
                    W[j+1,k]=W[k,j+1]=ww
                W[j+1,j]=W[j,j+1]=eps*normT/max(beta,1e-300); W[j+1,j+1]=1
                big=[k for k in range(j) if not abs(W[j+1,k])<=tau]
                if big:
                    if all(k<l for k in big): idx=[k for k in range(l) if abs(W[j+1,k])>tau*0.1]; nev_sel+=1
                    else: idx=list(range(j+1)); nev_full+=1
            if idx and beta>0:
                for p in range(2):
                    before=w@w
                    d=np.array([Bv[i]@w for i in idx]); reads+=2*len(idx)
                    cv=d*sc[idx]          # coefficient wrt normalised v_i
                    for t,i in enumerate(idx): w=w-cv[t]*sc[i]*Bv[i]
                    for t,i in enumerate(idx): F[i,j]+=cv[t]
                    if w@w>=0.5*before: break
                beta=np.linalg.norm(w)
                for k in idx: W[j+1,k]=W[k,j+1]=eps
            Bv[j+1]=w
            T[j+1,j]=T[j,j+1]=beta; beta_last=beta
            sc[j+1]=1/beta if beta>0 else 0
            if verbose:
                Q=np.array([sc[i]*Bv[i] for i in range(j+2)])
                G=Q@Q.T-np.eye(j+2)
                print(it,j,"true",np.abs(G[j+1,:j+1]).max(),"est",np.abs(W[j+1,:j+1]).max(),"idx",len(idx))
            uprev=u; s_prev=s_cur; beta_prev=beta
            j+=1
        Hs=T[:meff,:meff].copy(); theta,Y=np.linalg.eigh(Hs)
        want=min(nev,meff); nconv=0
        for i in range(want):
            resid=abs(beta_last*Y[meff-1,i])
            if resid<=tol_eff*max(eps23,abs(theta[i])): nconv+=1
        if nconv>=want or it+1>=maxiter: break
        lnew=min(m-1,nev+max(1,(m-nev)//2))
        newB=[sum(Bv[k]*sc[k]*Y[k,a] for k in range(m)) for a in range(lnew)]; reads+=m+lnew
        Wn=np.zeros((M,M))
        Wn[:lnew,:lnew]=Y[:,:lnew].T@W[:m,:m]@Y[:,:lnew]
        for a in range(lnew): Wn[a,a]=1
        Wn[lnew,:lnew]=W[m,:m]@Y[:,:lnew]; Wn[:lnew,lnew]=Wn[lnew,:lnew]; Wn[lnew,lnew]=1
        W=Wn
        # relation error of the new Ritz columns: old eta folded + dropped coefficients
        Fm=F[:m,:m]
        etan=np.zeros(M)
        for a in range(lnew):
            etan[a]=np.linalg.norm(Fm@Y[:,a])+np.abs(Y[:,a])@eta[:m]
        eta=etan; F[:]=0
        um=Bv[m]; scm=sc[m]
        for a in range(lnew): Bv[a]=newB[a]
        Bv[lnew]=um
        T[:]=0
        for a in range(lnew):
            T[a,a]=theta[a]; T[a,lnew]=T[lnew,a]=beta_last*Y[m-1,a]; sc[a]=1
        sc[lnew]=scm
        l=lnew; j=l
        uprev=Bv[l-1]; s_prev=0.0; beta_prev=0.0
    return theta[:nev], nmv, it+1, reads, nev_full, nev_sel
