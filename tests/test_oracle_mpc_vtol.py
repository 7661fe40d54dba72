"""CPU: the VTOL2D MPC-CBF problem functions (oracle/mpc_vtol.py) against the reference-executed fixture
tests/golden/mpc_functions.npz (f, g, x_next, step, the DT-CBF rows, MPCCBF's tables), and the forward-mode Jacobians against
finite differences.  No kernel serves VTOL2D yet and this repo's interior point does not converge on it:
what is held here is the problem statement a solver will be built against."""
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
