import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import safe_control_amd as sca
from safe_control_amd import workloads as W
dev="cuda:0"
for fam in ("du","uni","si","di","kb"):
    name=W.MPC_FAMILIES[fam]
    X,up,goal,obs=(a[:64].copy() for a in W.mpc_family_batch(fam,64,8,seed=1))
    X[0]=np.nan; X[1,0]=np.inf; up[2]=np.nan; goal[3]=np.inf; obs[4,0,0]=np.nan; obs[5,:,2]=-1.0; obs[6,:,:2]=X[6,:2]; X[7]=1e12; up[8]=1e9; obs[9,:,2]=1e6
    if X.shape[1]>=4: X[10,3]=50.0
    t=lambda a: torch.tensor(np.ascontiguousarray(a),dtype=torch.float64,device=dev)
    ctl=sca.BatchedMSMPCCBF({"model":name},io_dtype="f64",max_iter=300)
    t0=time.time(); u,st,it=ctl.solve(t(X),t(up),t(goal),t(obs)); torch.cuda.synchronize(); dt=time.time()-t0
    print(fam, "%.2f s"%dt, "status of the poisoned rows", st[:11].tolist(), "iterations", it[:11].tolist(), "others optimal", float((st[11:]==0).double().mean()))
