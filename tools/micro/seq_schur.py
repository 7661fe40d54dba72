import numpy as np, mpmath as mp, sys
mp.mp.dps=80
def run(S, e, C, Bm, gam, Mv0, Mr0, gv0, gr0, order=None):
    K=len(e)
    M=lambda a: mp.matrix(a.tolist())
    Kvv=M(Mv0); Kr=M(Mr0); D=M(S); gv=M(gv0.reshape(-1,1)); gr=M(gr0.reshape(-1,1))
    for j in range(K):
        c=M(C[j].reshape(-1,1)); b=M(Bm[j].reshape(-1,1)); ej=mp.mpf(e[j]); g=mp.mpf(gam[j])
        Kvv+=ej*c*c.T; Kr+=ej*b*c.T; D+=ej*b*b.T; gv+=g*c; gr+=g*b
    Dinv=D**-1
    f=lambda a: np.array(a.tolist(),dtype=float)
    Rex=f(Kvv-Kr.T*Dinv*Kr); gex=f(gv-Kr.T*Dinv*gr).ravel()
    Di=np.linalg.inv(S); T=Mr0.T@Di; R=Mv0-T@Mr0; g=gv0-T@gr0; t=Di@gr0
    idx = range(K) if order is None else order
    for j in idx:
        c=C[j]; b=Bm[j]
        w=Di@b; m=T@b; q=1/e[j]+b@w; r=c-m; sig=(gam[j]/e[j]-b@t)/q
        R+=np.outer(r,r)/q; g+=sig*r; T+=np.outer(r,w)/q; t+=sig*w; Di-=np.outer(w,w)/q
    Kvv=Mv0.copy(); Kr=Mr0.copy(); D=S.copy(); gv=gv0.copy(); gr=gr0.copy()
    for j in range(K):
        Kvv+=e[j]*np.outer(C[j],C[j]); Kr+=e[j]*np.outer(Bm[j],C[j]); D+=e[j]*np.outer(Bm[j],Bm[j]); gv+=gam[j]*C[j]; gr+=gam[j]*Bm[j]
    Rn=Kvv-Kr.T@np.linalg.inv(D)@Kr; gn=gv-Kr.T@np.linalg.inv(D)@gr
    sc=np.abs(Rex).max(); gs=np.abs(gex).max()
    return np.abs(R-Rex).max()/sc, np.abs(Rn-Rex).max()/sc, np.abs(g-gex).max()/gs, np.abs(gn-gex).max()/gs, sc
rng=np.random.default_rng(1)
for case in range(6):
    K=8
    Mv0=rng.normal(size=(6,6)); Mv0=1e-2*Mv0@Mv0.T
    Mr0=rng.normal(size=(2,6))*1e-2; l12=[0,0.005,0.05,0.5,0.0,0.005][case]
    S=np.array([[0.02,l12],[l12,0.02]])
    gv0=rng.normal(size=6)*1e-2; gr0=rng.normal(size=2)*1e-2
    C=rng.normal(size=(K,6))*30; Bm=rng.normal(size=(K,2))*50
    e=10**rng.uniform(-6,-2,K); nact=[1,1,1,1,2,0][case]
    e[:nact]=10**rng.uniform(8,11,nact)
    gam=e*rng.normal(size=K)
    print(case, 'l12',l12,'nact',nact, 'seq R %.1e naive R %.1e | seq g %.1e naive g %.1e scale %.1e'%run(S,e,C,Bm,gam,Mv0,Mr0,gv0,gr0))
