"""The reference's VTOL2D example (examples/test_vtol.py: 20 m/s start at (2, 10), waypoints (70, 10) and (70, 0.5), 24 known discs,
the 10 nearest unpassed ones per step as in tracking.py:345-404) flown by the lane solver of the HIP kernel compiled for the host
(tools/vtol_host.cpp): what statuses the interior point returns along the flight and whether the aircraft arrives.  CPU only,
a debugging aid.   [MAXIT=100] python tools/exp_vtol_closed_loop.py [steps]
Finding (round 3): the first NLPs of that example have no feasible point (the stage-0 CBF row cannot be met at 20 m/s, 58 m from the first
disc); the interior point ends its 100 iterations inside the restoration phase (status inaccurate) and the inputs of those unfinished
iterates pitch the aircraft past its 15 degree limit within 15 steps.  With a slack reset inside the restoration as well (tried, not
shipped) the first solve certifies infeasibility after 373 iterations at u0 = (1, 0, 0, -0.5), the least-violation input, and applying
THAT input passes the pitch limit after two steps.  What the reference flies on there is whatever IPOPT's own restoration returns."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dbg_vtol_host as Dh
from oracle import mpc_vtol as V
from oracle import mpc_cbf as M


def scene():
    p1, p2 = 67.0, 73.0
    obs = [[p1, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[p2, float(z), 0.5] for z in range(1, 16)] + [[60.0, 12.0, 1.5]]
    return np.array(obs), np.array([[2.0, 10.0], [70.0, 10.0], [70.0, 0.5]])


def nearest_unpassed(obs, x, num=10):
    pos, yaw = x[:2], x[2]
    ang = np.arctan2(obs[:, 1] - pos[1], obs[:, 0] - pos[0])
    diff = np.abs(((ang - yaw + np.pi) % (2 * np.pi)) - np.pi)
    front = obs[diff <= 0.6 * np.pi]
    pool = front if len(front) else obs
    d = np.linalg.norm(pool[:, :2] - pos, axis=1)
    return pool[np.argsort(d)[:num]]


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    lib = Dh.build()
    obs_all, wps = scene()
    spec = V.default_spec(radius=0.6, v_max=20.0)
    x = np.array([2.0, 10.0, 0.0, 20.0, 0.0, 0.0])
    up = np.zeros(4)
    gi = 0
    counts = np.zeros(3, dtype=int)
    for k in range(steps):
        if np.linalg.norm(x[:2] - wps[gi]) < 1.0:
            gi += 1
            if gi >= len(wps):
                print(f"step {k}: all waypoints reached"); break
        near = nearest_unpassed(obs_all, x)
        ob = M.pad_obstacles(near, 10)
        u, st, it, z = Dh.host_solve(lib, x, up, wps[gi], ob, v_max=20.0, max_iter=int(os.environ.get("MAXIT", "100")))
        counts[st] += 1
        x = V.vt_S(x, u, spec, 0.05)
        up = u
        clear = np.min(np.linalg.norm(obs_all[:, :2] - x[:2], axis=1) - obs_all[:, 2] - 0.6)
        if k % 10 == 0 or st != 0:
            print(f"{k:4d} goal {gi} status {st} it {it:3d} x {x[0]:6.2f} z {x[1]:5.2f} th {x[2]:+.3f} vx {x[3]:5.2f} vz {x[4]:+.2f} u {np.round(u, 3)} clearance {clear:.2f}", flush=True)
        if clear < 0 or x[1] < 0 or abs(x[2]) > np.radians(spec["pitch_max"]) * 1.0001 + 1e-9:
            print(f"step {k}: collision / ground / pitch limit"); break
    print("status counts (optimal, infeasible, inaccurate):", counts)
