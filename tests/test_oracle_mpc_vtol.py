"""CPU: the VTOL2D MPC-CBF problem functions (oracle/mpc_vtol.py) against the reference-executed fixture
tests/golden/mpc_functions.npz (f, g, x_next, step, the DT-CBF rows, MPCCBF's tables), the forward-mode first and second derivatives
against finite differences, and the two things that make the shared interior point converge on this model (round 3): the exact
Hessian of the aero model and the slack reset of the line search.  The kernel that serves VTOL2D is held to this oracle in
tests/test_mpcvtol_gpu.py; tests/test_vtol_solver_host.py runs the kernel's per-lane solver on the host against it."""
import os

import numpy as np

from oracle import mpc_gn as G
from oracle import mpc_vtol as V

D = np.load(os.path.join(os.path.dirname(__file__), "golden", "mpc_functions.npz"))
SPEC = V.default_spec(radius=float(D["VTOL2D/robot_radius"]))


def test_spec_defaults_are_the_reference_ones():
    ref = dict(zip([str(k) for k in D["VTOL2D/spec_keys"]], D["VTOL2D/spec_vals"]))
    for k, v in SPEC.items():
        assert k in ref and abs(ref[k] - v) <= 1e-15 * max(1.0, abs(v)), k


def test_dynamics_and_step():
    X, U = D["VTOL2D/x"], D["VTOL2D/u"]
    for i in range(len(X)):
        f, g = V.f_g_numeric(X[i], SPEC)
        assert np.abs(f - D["VTOL2D/f"][i]).max() <= 1e-12 * max(1.0, np.abs(f).max())
        assert np.abs(g - D["VTOL2D/g"][i]).max() <= 1e-12 * max(1.0, np.abs(g).max())
        assert np.abs(V.vt_F(X[i], U[i], SPEC, 0.05) - D["VTOL2D/x_next"][i]).max() <= 1e-12
        assert np.abs(V.vt_S(X[i], U[i], SPEC, 0.05) - D["VTOL2D/step"][i]).max() <= 1e-12


def test_forward_mode_jacobians():
    X, U = D["VTOL2D/x"], D["VTOL2D/u"]
    h = 1e-6
    for i in range(0, len(X), 4):
        _, A, B = V.vt_F(X[i], U[i], SPEC, 0.05, True)
        Afd = np.array([(V.vt_F(X[i] + h * e, U[i], SPEC, 0.05) - V.vt_F(X[i] - h * e, U[i], SPEC, 0.05)) / (2 * h) for e in np.eye(6)]).T
        Bfd = np.array([(V.vt_F(X[i], U[i] + h * e, SPEC, 0.05) - V.vt_F(X[i], U[i] - h * e, SPEC, 0.05)) / (2 * h) for e in np.eye(4)]).T
        assert np.abs(A - Afd).max() <= 1e-7 * max(1.0, np.abs(A).max()) and np.abs(B - Bfd).max() <= 1e-7 * max(1.0, np.abs(B).max())


def test_tables_and_cbf_rows():
    mdl = V.vtol_model(dict(radius=float(D["VTOL2D/robot_radius"])))
    assert np.array_equal(np.diag(D["VTOL2D/Q"]), mdl["Q"]) and np.array_equal(D["VTOL2D/R"], mdl["R"]) and int(D["VTOL2D/horizon"]) == 30
    assert (float(D["VTOL2D/cbf_param/alpha1"]), float(D["VTOL2D/cbf_param/alpha2"])) == (mdl["alpha1"], mdl["alpha2"])
    assert np.array_equal(D["VTOL2D/u_lo"], mdl["u_lo"]) and np.array_equal(D["VTOL2D/u_hi"], mdl["u_hi"])
    for idx, lo, hi in mdl["xb"]:
        assert D["VTOL2D/x_lo"][idx] == lo and D["VTOL2D/x_hi"][idx] == hi
    assert np.isinf(D["VTOL2D/x_lo"][[0, 1, 5]]).all() and np.isinf(D["VTOL2D/x_hi"][[0, 1, 4, 5]]).all()
    P = G.params(mdl, 1)
    X, U = D["VTOL2D/x"], D["VTOL2D/u"]
    for i in range(len(X)):
        ev = G.evaluate(X[i], U[i], np.zeros(4), D["VTOL2D/goal"][i], D["VTOL2D/obs"][i], P, level=1)
        K = D["VTOL2D/obs"][i].shape[0]
        ref = -D["VTOL2D/cons"][i]                           # the reference registers -(dd_h + (a1 + a2) d_h + a1 a2 h) <= 0
        assert np.abs(ev["g"][:K] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
        # rows: CBF | x_dot <= v_max, x_dot >= -v_max, z_dot >= -descent, |theta| <= pitch_max | input box
        assert ev["g"].shape[0] == K + 5 + 8 and ev["J"].shape == (K + 5 + 8, 4)


def test_the_reference_scene_starts_infeasible():
    """examples/test_vtol.py: x0 = (2, 10, 0, v_max = 20, 0, 0), one disc of radius 1.5 at (60, 12), robot radius 0.6.  The stage-0 CBF
    row against that disc depends on u_0 alone (x_0 is given) and is negative over the WHOLE input box: at 20 m/s and 58 m the
    0.1 d_h term (-11.6) outweighs 0.0025 h (+8.4) and two Euler steps of any admissible input move dd_h by a few tenths.  So the first
    NLP of the reference's own example has no feasible point; what the reference applies there is the output of IPOPT's restoration
    phase (MPCCBF.status is hard-wired to 'optimal', mpc_cbf.py:10,400) -- not reproducible without IPOPT."""
    import itertools
    from oracle import mpc_cbf as M
    mdl = V.vtol_model(dict(radius=0.6, v_max=20.0))
    x0 = np.array([2.0, 10.0, 0.0, 20.0, 0.0, 0.0])
    obs = np.array([[60.0, 12.0, 1.5, 0, 0, 0, 0]])
    P = G.params(mdl, 1)
    lo, hi = mdl["u_lo"], mdl["u_hi"]
    best = -np.inf
    for c in itertools.product(np.linspace(0, 1, 5), repeat=4):
        best = max(best, G.evaluate(x0, lo + (hi - lo) * np.array(c), np.zeros(4), np.array([70.0, 10.0]), obs, P, level=0)["g"][0])
    assert -1.0 < best < -0.7
    # the row is monotone enough in u_0 that the grid maximum sits on a vertex; a gradient check around it finds nothing better
    from scipy.optimize import minimize
    r = minimize(lambda u: -G.evaluate(x0, u, np.zeros(4), np.array([70.0, 10.0]), obs, P, level=0)["g"][0], hi, method="L-BFGS-B",
                 bounds=list(zip(lo, hi)))
    assert -r.fun <= best + 1e-9


def test_second_order_forward_mode_against_finite_differences_of_the_jacobian():
    X, U = D["VTOL2D/x"], D["VTOL2D/u"]
    rng = np.random.default_rng(0)
    h = 1e-5
    for i in range(0, len(X), 5):
        c = rng.normal(size=6)
        H = V.vt_H(X[i], U[i], SPEC, 0.05, c)

        def grad(z):
            _, A, B = V.vt_F(z[:6], z[6:], SPEC, 0.05, True)
            return c @ np.hstack([A, B])
        z = np.concatenate([X[i], U[i]])
        Hfd = np.array([(grad(z + h * e) - grad(z - h * e)) / (2 * h) for e in np.eye(10)])
        assert np.abs(H - H.T).max() <= 1e-12 * max(1.0, np.abs(H).max())
        assert np.abs(H - Hfd).max() <= 1e-5 * max(1.0, np.abs(H).max())
        assert np.abs(H[:2]).max() == 0.0 and np.abs(H[5, :6]).max() == 0.0    # positions and the pitch rate enter linearly


def test_feasible_probes_converge_with_exact_hessian_and_slack_reset():
    """Cruise with an obstacle 80 m ahead and a slow hover approach: optimal in a few dozen iterations; with the Gauss-Newton Hessian
    and the plain l1 line search the same solver ends its 100 iterations at KKT errors of 1e-2 .. 1e+2 (tools/exp_vtol.py)."""
    up = np.array([0.5, 0.5, 0.3, 0.0])
    probes = [(np.array([0, 10, 0.0, 12, 0, 0.0]), np.array([100.0, 10.0]), np.array([[80.0, 10.5, 1.5, 0, 0, 0, 0]])),
              (np.array([0, 10, 0.0, 2, 0, 0.0]), np.array([5.0, 10.0]), np.array([[1e3, 1e3, 0.0, 0, 0, 0, 0]]))]
    for x0, goal, obs in probes:
        u, st, it, info = V.solve(x0, up, goal, obs, return_info=True)
        assert st == 0 and it <= 40 and info["err"] <= 1e-6 and info["theta"] == 0.0
        mdl = V.vtol_model()
        assert np.all(u >= mdl["u_lo"] - 1e-9) and np.all(u <= mdl["u_hi"] + 1e-9)
    x0, goal, obs = probes[0]
    _, st, it, info = V.solve(x0, up, goal, obs, params_over=dict(slack_reset=0, exact_hessian=False, max_iter=60), return_info=True)
    assert st != 0 and info["err"] > 1e-3


def test_slack_reset_leaves_a_converging_family_where_it_was():
    """The reset only changes which trial steps the line search accepts: on DynamicUnicycle2D problems that converge anyway the
    minimiser is the same to solver precision."""
    from oracle import mpc_cbf as M
    from safe_control_amd import workloads as W
    X, goal, _, obs = W.du_cbfqp_batch(8, 8, seed=0)
    P = dict(M.DEFAULTS, a_max=1.0, w_max=0.5)
    for i in range(8):
        u0, s0, _ = M.solve(X[i], np.zeros(2), goal[i], obs[i], params=P)
        u2, s2, _ = M.solve(X[i], np.zeros(2), goal[i], obs[i], params=dict(P, slack_reset=2))
        if s0 == 0 and s2 == 0:
            assert np.abs(u0 - u2).max() <= 1e-6
