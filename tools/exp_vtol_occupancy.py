import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
import safe_control_amd as sca
from safe_control_amd import workloads as W
X,up,goal,obs=W.mpc_family_batch("vtol",4096,8,seed=0)
tt=lambda a: torch.tensor(np.ascontiguousarray(a),dtype=torch.float64,device="cuda:0")
ctl=sca.BatchedVtolMPCCBF(io_dtype="f64")
for B in (256,512,768,1024,1280,1536,2048,4096):
    rep=lambda a: tt(np.repeat(a[:1],B,axis=0))
    a=(rep(X),rep(up),rep(goal),rep(obs))
    ctl.solve(*a); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); u,st,it=ctl.solve(*a); e1.record(); torch.cuda.synchronize()
    print(B, f"{e0.elapsed_time(e1):.2f} ms", int(it[0]), flush=True)
