"""Slack reset in the line search of the MPC interior point (oracle, CPU): iterations and statuses of the first n problems of each
family's bench batch with P["slack_reset"] = 0 / 1 / 2.   python tools/exp_slack_reset.py n family [family ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_control_amd import workloads as W
from tests._oracle_pool import family_solve_many

if __name__ == "__main__":
    n = int(sys.argv[1]); fams = sys.argv[2:]
    modes = [int(m) for m in os.environ.get("MODES", "0,2").split(",")]
    for fam in fams:
        X, up, goal, obs = W.mpc_family_batch(fam, 4096, 8, seed=0)[:4]
        X, up, goal, obs = X[:n], up[:n], goal[:n], obs[:n]
        ref = None
        for mode in modes:
            t = time.time()
            r = family_solve_many(fam, X, up, goal, obs, params=dict(slack_reset=mode % 10, resto_slack_reset=mode >= 10), workers=8, timeout=3000)
            st, it = r["st"], r["it"]
            line = f"{fam:7s} reset {mode}: opt {np.mean(st == 0):.3f} inf {np.mean(st == 1):.3f} inacc {np.mean(st == 2):.3f} | it mean {it.mean():.1f} p95 {np.percentile(it, 95):.0f} max {it.max()} n_resto {np.mean(r['n_resto'] > 0):.3f}"
            if ref is not None:
                both = (ref["st"] == 0) & (st == 0)
                line += f" | both optimal {both.mean():.3f}, max |du0| among them {np.abs(ref['u'] - r['u'])[both].max():.2e}"
            else:
                ref = r
            print(line + f"  ({time.time() - t:.0f}s)", flush=True)
