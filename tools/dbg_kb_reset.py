import sys, os, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import safe_control_amd as sca
from safe_control_amd import workloads as W
from _oracle_pool import family_solve_many
n=64
for fam in ("kb","c3bf"):
    X,up,goal,obs=(a[:n] for a in W.mpc_family_batch(fam,4096,8,seed=0))
    t=lambda a: torch.tensor(np.ascontiguousarray(a),dtype=torch.float64,device="cuda:0")
    for sr in (0,2):
        ctl=sca.BatchedGnMPCCBF({"model":W.MPC_FAMILIES[fam]},io_dtype="f64")
        ctl._mc["slack_reset"]=sr
        u,st,it,z=ctl.solve(t(X),t(up),t(goal),t(obs),want_z=True); torch.cuda.synchronize()
        o=family_solve_many(fam,X,up,goal,obs,params=dict(slack_reset=sr))
        st=st.cpu().numpy(); it=it.cpu().numpy()
        print(fam,"slack_reset",sr,"gpu status",np.bincount(st,minlength=3),"oracle",np.bincount(o['st'],minlength=3),"iters equal",np.mean(it==o['it']),"gpu it mean",it.mean(),"oracle",o['it'].mean(), flush=True)
