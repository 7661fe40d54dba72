"""Seeded synthetic workloads for BASELINE.json's configurations.

Vectorised numpy generation of agent states, goals, nominal inputs and
obstacle tables.  This prepares *inputs* for tests and bench.py; it is not on
the solve path.
"""
import math

import numpy as np


def wrap_angle(x):
    """Python-% wrap into [-pi, pi) (robots/dynamic_unicycle2D.py:13-16)."""
    return np.mod(x + np.pi, 2.0 * np.pi) - np.pi


def nominal_input_du(X, goal, v_max=1.0, k_omega=2.0, k_a=1.0, k_v=1.0, d_min=0.05):
    """Vectorised DynamicUnicycle2D.nominal_input (robots/dynamic_unicycle2D.py:80-104)."""
    d = np.maximum(np.hypot(X[:, 0] - goal[:, 0], X[:, 1] - goal[:, 1]) - d_min, 0.0)
    err = wrap_angle(np.arctan2(goal[:, 1] - X[:, 1], goal[:, 0] - X[:, 0]) - X[:, 2])
    v = np.where(np.abs(err) > math.radians(90), 0.0, np.minimum(k_v * d * np.cos(err), v_max))
    return np.stack([k_a * (v - X[:, 3]), k_omega * err], axis=1)


def nominal_input_kb(X, goal, spec, k_theta=2.0, k_a=1.0, k_v=1.0, d_min=0.05):
    """Vectorised KinematicBicycle2D.nominal_input with the gains BaseRobot forwards
    (robots/kinematic_bicycle2D.py:125-147 via robots/robot.py:401-408)."""
    d = np.maximum(np.hypot(X[:, 0] - goal[:, 0], X[:, 1] - goal[:, 1]) - d_min, 0.05)
    err = wrap_angle(np.arctan2(goal[:, 1] - X[:, 1], goal[:, 0] - X[:, 0]) - X[:, 2])
    delta = np.clip(k_theta * err, -spec["delta_max"], spec["delta_max"])
    beta = np.arctan((spec["rear_ax_dist"] / spec["wheel_base"]) * np.tan(delta))
    v = np.clip(k_v * d * np.maximum(0.0, np.cos(err)), spec["v_min"], spec["v_max"])
    return np.stack([k_a * (v - X[:, 3]), beta], axis=1)


def du_cbfqp_batch(B=4096, K=8, seed=0, radius=0.25, v_max=1.0):
    """BASELINE config 2: B DynamicUnicycle2D agents, K circular obstacles each.

    x,y ~ U(0,14); theta ~ U(-pi,pi); v ~ U(0,1); goal ~ U(0,14)^2; u_ref from the
    nominal controller; obstacle k of agent i: r ~ U(.2,1), centre at polar offset
    rho ~ U(r+R+.05, 4), phi ~ U(-pi,pi) from the agent; columns 3..6 zero (circle).
    Returns float64 arrays X[B,4], goal[B,2], u_ref[B,2], obs[B,K,7].
    """
    rng = np.random.default_rng(seed)
    X = np.empty((B, 4))
    X[:, 0:2] = rng.uniform(0.0, 14.0, (B, 2))
    X[:, 2] = rng.uniform(-np.pi, np.pi, B)
    X[:, 3] = rng.uniform(0.0, 1.0, B)
    goal = rng.uniform(0.0, 14.0, (B, 2))
    r = rng.uniform(0.2, 1.0, (B, K))
    rho = rng.uniform(0.0, 1.0, (B, K)) * (4.0 - (r + radius + 0.05)) + (r + radius + 0.05)
    phi = rng.uniform(-np.pi, np.pi, (B, K))
    obs = np.zeros((B, K, 7))
    obs[:, :, 0] = X[:, None, 0] + rho * np.cos(phi)
    obs[:, :, 1] = X[:, None, 1] + rho * np.sin(phi)
    obs[:, :, 2] = r
    u_ref = nominal_input_du(X, goal, v_max=v_max)
    return X, goal, u_ref, obs


def kb_c3bf_batch(B=16384, K=16, seed=0, spec=None, shared_table=False):
    """BASELINE config 4: KinematicBicycle2D (C3BF/DPCBF) agents with K moving circles.

    v ~ U(.2,3.5); obstacles r=.5, vx,vy ~ U(-.5,.5), placed at polar offset
    rho ~ U(r+R+.3, 8) from the agent (or uniformly in the 14x14 field when one
    table is shared by all agents).
    """
    from .robots.spec import complete_robot_spec
    spec = complete_robot_spec(dict(spec or {"model": "KinematicBicycle2D_C3BF"}))
    rng = np.random.default_rng(seed)
    X = np.empty((B, 4))
    X[:, 0:2] = rng.uniform(0.0, 14.0, (B, 2))
    X[:, 2] = rng.uniform(-np.pi, np.pi, B)
    X[:, 3] = rng.uniform(0.2, 3.5, B)
    goal = rng.uniform(0.0, 14.0, (B, 2))
    R = spec["radius"]
    if shared_table:
        obs = np.zeros((K, 7))
        obs[:, 0:2] = rng.uniform(0.0, 14.0, (K, 2))
        obs[:, 2] = 0.5
        obs[:, 3:5] = rng.uniform(-0.5, 0.5, (K, 2))
    else:
        rho = rng.uniform(0.5 + R + 0.3, 8.0, (B, K))
        phi = rng.uniform(-np.pi, np.pi, (B, K))
        obs = np.zeros((B, K, 7))
        obs[:, :, 0] = X[:, None, 0] + rho * np.cos(phi)
        obs[:, :, 1] = X[:, None, 1] + rho * np.sin(phi)
        obs[:, :, 2] = 0.5
        obs[:, :, 3:5] = rng.uniform(-0.5, 0.5, (B, K, 2))
    u_ref = nominal_input_kb(X, goal, spec)
    return X, goal, u_ref, obs


def linear_mpc_batch(model="Quad3D", B=4096, K=8, seed=0, radius=0.25):
    """Batch for the linear-model MPC-CBF kernel (BASELINE config 5 names Quad3D): positions ~ U(0,14)^2, goals
    ~ U(0,14)^2 (Quad3D: altitude 1..2, goal altitude 1..2, planar speed up to 0.8 m/s, small attitude), K circular
    obstacles per agent as in du_cbfqp_batch.  Returns float64 X[B,nx], goal[B,ng], obs[B,K,7]."""
    rng = np.random.default_rng(seed)
    nx, ng = (12, 3) if model == "Quad3D" else (2, 2)
    X = np.zeros((B, nx))
    X[:, 0:2] = rng.uniform(0.0, 14.0, (B, 2))
    goal = np.zeros((B, ng))
    goal[:, 0:2] = rng.uniform(0.0, 14.0, (B, 2))
    if model == "Quad3D":
        X[:, 2] = rng.uniform(1.0, 2.0, B)
        X[:, 3:5] = rng.uniform(-0.05, 0.05, (B, 2))
        X[:, 6:8] = rng.uniform(-0.8, 0.8, (B, 2)) / np.sqrt(2.0)
        goal[:, 2] = rng.uniform(1.0, 2.0, B)
    r = rng.uniform(0.2, 1.0, (B, K))
    rho = rng.uniform(0.0, 1.0, (B, K)) * (4.0 - (r + radius + 0.05)) + (r + radius + 0.05)
    phi = rng.uniform(-np.pi, np.pi, (B, K))
    obs = np.zeros((B, K, 7))
    obs[:, :, 0] = X[:, None, 0] + rho * np.cos(phi)
    obs[:, :, 1] = X[:, None, 1] + rho * np.sin(phi)
    obs[:, :, 2] = r
    return X, goal, obs


MPC_FAMILIES = {"du": "DynamicUnicycle2D", "kb": "KinematicBicycle2D", "c3bf": "KinematicBicycle2D_C3BF", "dpcbf": "KinematicBicycle2D_DPCBF",
                "di": "DoubleIntegrator2D", "quad2d": "Quad2D", "si": "SingleIntegrator2D", "quad3d": "Quad3D", "vtol": "VTOL2D", "uni": "Unicycle2D"}


def mpc_family_batch(family, B=4096, K=8, seed=0):
    """The MPC-CBF batches bench.py times and the full-batch parity tests solve, one per model family (keys of MPC_FAMILIES):
    positions, goals and K circles per agent from du_cbfqp_batch(seed) (BASELINE config 3's draws) -- linear_mpc_batch(seed) for
    the two linear models -- with the model's remaining states from default_rng(seed + 1): bicycles drive roughly towards their
    goal at 0.5 .. 3 m/s, DoubleIntegrator2D velocities in +-0.7 m/s, Quad2D near hover.  Returns float64 X, u_prev, goal, obs."""
    from .robots.spec import complete_robot_spec
    if family == "vtol":
        return vtol_mpc_batch(B, K, seed=seed)
    if family in ("si", "quad3d"):
        X, goal, obs = linear_mpc_batch(MPC_FAMILIES[family], B, K, seed=seed)
        return X, np.zeros((B, 4 if family == "quad3d" else 2)), goal, obs
    Xd, goal, _, obs = du_cbfqp_batch(B, K, seed=seed)
    rng = np.random.default_rng(seed + 1)
    up = np.zeros((B, 2))
    if family == "du":
        X = Xd
    elif family == "quad2d":
        spec = complete_robot_spec({"model": "Quad2D"})
        X = np.zeros((B, 6)); X[:, 0:2] = Xd[:, 0:2]; X[:, 2] = rng.uniform(-0.2, 0.2, B); X[:, 3:5] = rng.uniform(-0.5, 0.5, (B, 2))
        up = np.full((B, 2), 0.5 * (spec["f_min"] + spec["f_max"]))
    elif family in ("kb", "c3bf", "dpcbf"):
        X = np.zeros((B, 4)); X[:, 0:2] = Xd[:, 0:2]
        X[:, 2] = np.arctan2(goal[:, 1] - Xd[:, 1], goal[:, 0] - Xd[:, 0]) + rng.uniform(-0.6, 0.6, B); X[:, 3] = rng.uniform(0.5, 3.0, B)
    elif family == "di":
        X = np.zeros((B, 4)); X[:, 0:2] = Xd[:, 0:2]; X[:, 2:4] = rng.uniform(-0.7, 0.7, (B, 2))
    elif family == "uni":                                   # Unicycle2D: (x, y, theta) padded to four columns; last input (v, w) as the DynamicUnicycle2D draw's (speed, 0)
        X = np.zeros((B, 4)); X[:, 0:3] = Xd[:, 0:3]
        up[:, 0] = Xd[:, 3]
    else:
        raise KeyError(family)
    return X, up, goal, obs


def vtol_mpc_batch(B=4096, K=8, seed=0):
    """VTOL2D MPC-CBF batch (x-z plane; examples/test_vtol.py's flight regime scaled to where the first NLP is feasible): cruise at
    8 .. 14 m/s, altitude 8 .. 12 m, pitch and pitch rate near zero, a goal 60 .. 120 m ahead within +-3 m of altitude, K discs of
    radius 0.5 .. 2 m placed 60 .. 150 m away inside a +-30 degree cone ahead (the stage-0 DT-CBF row with alpha = 0.05 needs roughly
    distance >= 4 x closing speed), previous input near trim.  Returns float64 X[B,6], u_prev[B,4], goal[B,2], obs[B,K,7]."""
    rng = np.random.default_rng(seed)
    X = np.zeros((B, 6))
    X[:, 0] = rng.uniform(-5.0, 5.0, B); X[:, 1] = rng.uniform(8.0, 12.0, B)
    X[:, 2] = rng.uniform(-0.05, 0.05, B); X[:, 3] = rng.uniform(8.0, 14.0, B)
    X[:, 4] = rng.uniform(-1.0, 1.0, B); X[:, 5] = rng.uniform(-0.05, 0.05, B)
    goal = np.stack([X[:, 0] + rng.uniform(60.0, 120.0, B), X[:, 1] + rng.uniform(-3.0, 3.0, B)], axis=1)
    up = np.stack([rng.uniform(0.4, 0.6, B), rng.uniform(0.4, 0.6, B), rng.uniform(0.2, 0.5, B), rng.uniform(-0.02, 0.02, B)], axis=1)
    d = rng.uniform(60.0, 150.0, (B, K)); phi = rng.uniform(-np.pi / 6, np.pi / 6, (B, K))
    obs = np.zeros((B, K, 7))
    obs[..., 0] = X[:, None, 0] + d * np.cos(phi)
    obs[..., 1] = X[:, None, 1] + d * np.sin(phi)
    obs[..., 2] = rng.uniform(0.5, 2.0, (B, K))
    return X, up, goal, obs


def superellipsoid_obstacles(pos, K=8, seed=0, radius=0.25, exponents=(4.0, 6.0), rho_max=4.0):
    """BASELINE config 5's obstacles: K superellipsoid rows ``[ox, oy, a, b, e, theta, 1]`` per agent (the 7-wide layout of
    robots/dynamic_unicycle2D.py:148-183 / :204-220), semi-axes U(0.2, 0.8), exponent drawn from ``exponents``, random
    orientation, centred at a distance that keeps the inflated shape clear of the agent at the start (h > 0)."""
    rng = np.random.default_rng(seed)
    pos = np.asarray(pos, dtype=np.float64)
    B = pos.shape[0]
    a = rng.uniform(0.2, 0.8, (B, K)); b = rng.uniform(0.2, 0.8, (B, K))
    e = rng.choice(np.asarray(exponents, dtype=np.float64), (B, K))
    th = rng.uniform(-np.pi, np.pi, (B, K))
    clear = np.hypot(a + radius, b + radius) + 0.05                       # the corner of the bounding box of the inflated shape
    rho = clear + rng.uniform(0.0, 1.0, (B, K)) * np.maximum(rho_max - clear, 0.1)
    phi = rng.uniform(-np.pi, np.pi, (B, K))
    obs = np.zeros((B, K, 7))
    obs[..., 0] = pos[:, None, 0] + rho * np.cos(phi)
    obs[..., 1] = pos[:, None, 1] + rho * np.sin(phi)
    obs[..., 2], obs[..., 3], obs[..., 4], obs[..., 5], obs[..., 6] = a, b, e, th, 1.0
    return obs
