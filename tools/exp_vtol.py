"""VTOL2D MPC-CBF probes on the numpy oracle: does the shared interior point converge on feasible VTOL2D problems, with the
Gauss-Newton Hessian and with the exact one (second-order forward mode through the aero model)?  CPU only.
  python tools/exp_vtol.py [exact|gn] [max_iter]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpc_vtol as V

FAR = np.array([[1e4, 1e4, 0.1, 0, 0, 0, 0]])
PROBES = {
    "cruise12": (np.array([0, 10, 0.0, 12, 0, 0.0]), np.array([60.0, 10.0]), FAR),
    "cruise15": (np.array([0, 10, 0.0, 15, 0, 0.0]), np.array([60.0, 10.0]), FAR),
    "hover2": (np.array([0, 10, 0.0, 2, 0, 0.0]), np.array([5.0, 10.0]), FAR),
    "obst80": (np.array([0, 10, 0.0, 12, 0, 0.0]), np.array([100.0, 10.0]), np.array([[80.0, 10.5, 1.5, 0, 0, 0, 0]])),
    "climb": (np.array([0, 10, 0.05, 10, 1.0, 0.0]), np.array([40.0, 14.0]), FAR),
}

if __name__ == "__main__":
    exact = (sys.argv[1] if len(sys.argv) > 1 else "exact") == "exact"
    mi = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    only = sys.argv[3:] or list(PROBES)
    for name in only:
        x0, goal, obs = PROBES[name]
        t = time.time()
        u, st, it, info = V.solve(x0, np.array([0.5, 0.5, 0.3, 0.0]), goal, obs, params_over=dict(exact_hessian=exact, max_iter=mi, slack_reset=int(os.environ.get("SRESET", "0"))),
                                  return_info=True)
        print(f"{name:9s} exact={exact} status {st} it {it} err {info['err']:.2e} theta {info['theta']:.1e} n_resto {info['n_resto']} "
              f"f {info['f']:.4f} u0 {np.round(u, 4)}  {time.time() - t:.0f}s", flush=True)
