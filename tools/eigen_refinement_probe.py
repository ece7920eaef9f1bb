"""Would an MFMA-friendly refinement of a warm eigenbasis (Ogita & Aishima 2018: X <- X (I + E), four 64^3 products per step) replace the
Jacobi sweeps of eigen_omega_kernel's warm starts?  numpy, no GPU: kappa/omega rate matrices with random F3x4 frequencies, the eigenbasis at
kappa0 refined towards kappa0 (1 + d).  Output per step: off-diagonal mass of X^T B X relative to the diagonal / ||I - X^T X|| (pairs treated
as a cluster).  Result: converges in 3-4 steps only for d <= 1e-3; at 1 % it crawls and from 5 % on it stalls -- the
spectrum's gaps (minimum 2e-3, median 2e-2 of a norm of 8) are smaller than the perturbation, and the method's cluster rule then refines
nothing.  The kappa probes of OMEGA move by per cents.  Not a replacement.   python tools/eigen_refinement_probe.py"""
import numpy as np
rng=np.random.default_rng(1)
codons=[a+b+c for a in "TCAG" for b in "TCAG" for c in "TCAG"]
aa="FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
# the kernel's codon order is ACGT-based (index = 16a+4b+c with A,C,G,T = 0..3); use that order
nuc="ACGT"
cod=[a+b+c for a in nuc for b in nuc for c in nuc]
aa_of={c:aa[codons.index(c)] for c in cod}
def B_of(kappa,omega,pi):
    Q=np.zeros((64,64))
    for i,ci in enumerate(cod):
        for j,cj in enumerate(cod):
            d=[k for k in range(3) if ci[k]!=cj[k]]
            if len(d)!=1: continue
            a,b=ci[d[0]],cj[d[0]]
            ts={a,b} in ({"A","G"},{"C","T"})
            r=kappa if ts else 1.0
            ai,aj=aa_of[ci],aa_of[cj]
            if ai!='*' and aj!='*' and ai!=aj: r*=omega
            Q[i,j]=r*pi[j]
    np.fill_diagonal(Q,-Q.sum(1))
    norm=-(pi*np.diag(Q)).sum()
    Q/=norm
    root=np.sqrt(pi)
    Bm=root[:,None]*Q/root[None,:]
    return 0.5*(Bm+Bm.T)
def oa_step(A,X):
    n=64
    R=np.eye(n)-X.T@X
    S=X.T@(A@X)
    lam=np.diag(S)/(1-np.diag(R))
    D=np.diag(lam)
    delta=2*(np.linalg.norm(S-D,'fro')+np.linalg.norm(A,'fro')*np.linalg.norm(R,'fro'))
    E=np.empty((n,n))
    gap=lam[None,:]-lam[:,None]
    far=np.abs(gap)>delta
    with np.errstate(divide='ignore',invalid='ignore'):
        E=np.where(far,(S+lam[None,:]*R)/gap,R/2)
    return X+X@E, np.linalg.norm(E,'fro'), int((~far).sum()-n)
def offdiag(A,X):
    S=X.T@A@X
    off=np.linalg.norm(S-np.diag(np.diag(S)),'fro'); dg=np.linalg.norm(np.diag(S))
    return off/dg, np.linalg.norm(np.eye(64)-X.T@X,'fro')
for trial in range(3):
    f=rng.dirichlet(np.ones(4)*5,size=3)
    pi=np.array([f[0][nuc.index(c[0])]*f[1][nuc.index(c[1])]*f[2][nuc.index(c[2])] for c in cod]); pi/=pi.sum()
    k0=float(rng.uniform(1.5,4)); om=float(rng.uniform(0.05,1.0))
    w,X0=np.linalg.eigh(B_of(k0,om,pi))
    gaps=np.diff(np.sort(w)); print("trial",trial,"kappa0 %.2f omega %.2f"%(k0,om),"min gap %.2e"%gaps.min(),"median gap %.2e"%np.median(gaps),"norm %.2f"%np.linalg.norm(B_of(k0,om,pi),'fro'))
    for d in (1e-4,1e-3,1e-2,0.05,0.1,0.3,1.0):
        A=B_of(k0*(1+d),om,pi)
        X=X0.copy(); hist=[]
        for it in range(8):
            X,e,ncl=oa_step(A,X)
            o,r=offdiag(A,X)
            hist.append("%.0e/%.0e(%d)"%(o,r,ncl))
            if o<1e-15 and r<1e-14: break
        print("   dk/k %-6g start off %.1e -> %s"%(d,offdiag(A,X0)[0]," ".join(hist)))
