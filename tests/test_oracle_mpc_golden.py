"""The MPC-CBF PROBLEM pinned on the reference's own code.

tests/golden/mpc_functions.npz was recorded by tests/golden/make_golden.py::gen_mpc_functions, which EXECUTES the
reference's MPCCBF.__init__ / create_model / create_mpc / set_cbf_constraint / update_tvp and every robot's
agent_barrier_dt / f_casadi / g_casadi (position_control/mpc_cbf.py, robots/*.py) under a numeric casadi + a do_mpc
recorder.  Here the oracles' problem functions and the product's host-side tables are held to those values at 1e-12
(relative).  What remains unpinned after this file: IPOPT's choice of local optimum (no solver in the image).
"""
import os

import numpy as np
import pytest

from oracle import mpc_cbf as M
from oracle import mpc_cbf_uni as MU
from oracle import mpc_gn as G
from oracle import mpc_kb_state as KS
from oracle import mpc_lin as L

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mpc_functions.npz"))
RTOL = 1e-12


def fx(name, key):
    return GOLD[f"{name}/{key}"]


def spec_of(name):
    return dict(zip([str(k) for k in fx(name, "spec_keys")], [float(v) for v in fx(name, "spec_vals")]))


def close(a, b, rtol=RTOL, scale=1.0):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.all(np.abs(a - b) <= rtol * np.maximum(scale, np.maximum(np.abs(a), np.abs(b))))


def one_stage(name):
    """(evaluate(x, u, goal, obs) -> dict(f, g, X)) of the model's oracle with a one-step horizon and u_prev = 0."""
    sp = spec_of(name)
    R = float(fx(name, "robot_radius"))
    if name == "DynamicUnicycle2D":
        P = dict(M.DEFAULTS, N=1, radius=R, v_max=sp["v_max"], a_max=sp["a_max"], w_max=sp["w_max"])
        return lambda x, u, goal, obs: M.evaluate(x, u, np.zeros(2), goal, obs, P, level=0), P
    if name == "Unicycle2D":
        P = dict(MU.DEFAULTS, N=1, radius=R)
        return lambda x, u, goal, obs: MU.evaluate(x, u, np.zeros(2), goal, obs, P, level=0), P
    if name in ("SingleIntegrator2D", "Quad3D"):
        mdl = L.si_model(dict(v_max=sp["v_max"], radius=R)) if name == "SingleIntegrator2D" else L.quad3d_model(dict(sp, radius=R))
        P = L.params(mdl, N=1)
        return lambda x, u, goal, obs: L.evaluate(x, u, np.zeros(mdl["nu"]), goal, obs, P, level=0), P
    if name in ("KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"):     # full-state DT barriers, one gain
        mdl = (KS.c3bf_model if name.endswith("C3BF") else KS.dpcbf_model)(dict(sp, radius=R))
        P = KS.params(mdl, N=1)
        return lambda x, u, goal, obs: KS.evaluate(x, u, np.zeros(2), goal, obs, P, level=0), P
    mk = {"DoubleIntegrator2D": G.di_model, "Quad2D": G.quad2d_model, "KinematicBicycle2D": G.kb_model}[name]
    mdl = mk(dict(sp, radius=R))
    P = G.params(mdl, N=1)
    return lambda x, u, goal, obs: G.evaluate(x, u, np.zeros(2), goal, obs, P, level=0), P


ORACLE_MODELS = ["DynamicUnicycle2D", "Unicycle2D", "SingleIntegrator2D", "Quad3D", "DoubleIntegrator2D", "Quad2D",
                 "KinematicBicycle2D", "KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"]


@pytest.mark.parametrize("name", ORACLE_MODELS)
def test_cbf_rows_prediction_and_cost_equal_the_reference(name):
    """g[:K] of a one-stage problem is the reference's registered constraint (mpc_cbf.py:304: -cbf <= 0), the predicted
    state is its x_next (:138-141) and f is its stage cost at x_next (:144) plus the rterm of do-mpc (:180)."""
    ev, P = one_stage(name)
    x, u, goal, obs = fx(name, "x"), fx(name, "u"), fx(name, "goal"), fx(name, "obs")
    cons, xn, cn = fx(name, "cons"), fx(name, "x_next"), fx(name, "cost_next")
    Rw = fx(name, "rterm_u")
    K = obs.shape[1]
    for i in range(x.shape[0]):
        r = ev(x[i], u[i], goal[i], obs[i])
        big = np.maximum(1.0, np.abs(obs[i][:, 0:2]).max() ** 2)        # the dummy row's h is ~2e6: relative to that
        assert close(r["g"][:K], -cons[i], scale=1.0) or close(r["g"][:K], -cons[i], rtol=1e-15, scale=big), (name, i, r["g"][:K], -cons[i])
        assert close(r["X"][1], xn[i]), (name, i)
        assert close(r["f"], cn[i] + float(np.sum(Rw * u[i] ** 2)), rtol=1e-11), (name, i)


@pytest.mark.parametrize("name", ORACLE_MODELS)
def test_barrier_differences_equal_agent_barrier_dt(name):
    """(h_k, d_h, dd_h) returned by the reference's agent_barrier_dt recombine to the registered row:
    dd_h + (a1 + a2) d_h + a1 a2 h_k (mpc_cbf.py:316-321) or d_h + a h_k (:312-315)."""
    hk, dh, ddh, cons = fx(name, "hk"), fx(name, "dh"), fx(name, "ddh"), fx(name, "cons")
    if np.isnan(ddh).all():
        a = float(fx(name, "cbf_param/alpha"))
        row = dh + a * hk
    else:
        a1, a2 = float(fx(name, "cbf_param/alpha1")), float(fx(name, "cbf_param/alpha2"))
        row = ddh + (a1 + a2) * dh + a1 * a2 * hk
    assert close(row, -cons, rtol=1e-13, scale=np.abs(hk).max())


def test_host_tables_equal_the_reference_tables():
    """Q, R, horizon, gains, n_states / n_controls and the input / state boxes of the product's host classes against
    MPCCBF.__init__ / create_mpc of the reference (mpc_cbf.py:15-95, :182-232)."""
    from safe_control_amd.position_control import mpc_cbf as H
    from safe_control_amd.position_control import mpc_cbf_gn as HG
    from safe_control_amd.robots import linear_models as HL
    from safe_control_amd.robots.spec import complete_robot_spec

    for name in ("DynamicUnicycle2D", "Unicycle2D"):
        Q, R = H.default_mpc_weights(name)
        assert np.array_equal(Q, fx(name, "Q")) and np.array_equal(R, fx(name, "R"))
        cp = H.default_mpc_cbf_param(name)
        for k, v in cp.items():
            assert v == float(fx(name, f"cbf_param/{k}"))
        assert sorted(cp) == sorted(k.split("/")[-1] for k in GOLD.files if k.startswith(f"{name}/cbf_param/"))
    for name in ("SingleIntegrator2D", "Quad3D"):
        sp = complete_robot_spec(dict(spec_of(name), model=name))
        mdl = HL.linear_model(sp, 0.05)
        assert np.array_equal(mdl["Q"], fx(name, "Q")) and np.array_equal(mdl["R"], fx(name, "R"))
        assert mdl["cbf_param"]["alpha"] == float(fx(name, "cbf_param/alpha"))
        assert np.array_equal(mdl["u_lo"], fx(name, "u_lo")) and np.array_equal(mdl["u_hi"], fx(name, "u_hi"))
        assert mdl["nx"] == int(fx(name, "n_states")) and mdl["nu"] == int(fx(name, "n_controls"))
        # prediction model x + (f + g u) dt of create_model against Ae, Be
        x, u, xn = fx(name, "x"), fx(name, "u"), fx(name, "x_next")
        assert close(x @ mdl["Ae"].T + u @ mdl["Be"].T, xn)
        # the barrier's own step (Euler for SI, RK4 for Quad3D) against As, Bs
        st = fx(name, "step")
        assert close(x @ mdl["As"].T + u @ mdl["Bs"].T, st, rtol=1e-11)
    for name in ("KinematicBicycle2D", "KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"):
        sp = complete_robot_spec(dict(spec_of(name), model=name))
        mc = HG.model_constants(sp)
        assert np.array_equal(np.diag(mc["Q"]), fx(name, "Q")) and np.array_equal(mc["R"], fx(name, "R"))
        assert mc["cbf_param"] == {k.split("/")[-1]: float(GOLD[k]) for k in GOLD.files if k.startswith(f"{name}/cbf_param/")}
        assert close(mc["u_lo"], fx(name, "u_lo")) and close(mc["u_hi"], fx(name, "u_hi")) and mc["nx"] == int(fx(name, "n_states"))
        xlo, xhi = fx(name, "x_lo"), fx(name, "x_hi")                    # the speed is the one bounded state (mpc_cbf.py:205-207)
        assert np.isinf(xlo[:3]).all() and np.isinf(xhi[:3]).all() and (xlo[3], xhi[3]) == (-sp["v_max"], sp["v_max"])
    for name in ("DoubleIntegrator2D", "Quad2D"):
        sp = complete_robot_spec(dict(spec_of(name), model=name))
        mc = HG.model_constants(sp)
        assert np.array_equal(np.diag(mc["Q"]), fx(name, "Q")) and np.array_equal(mc["R"], fx(name, "R"))
        assert mc["cbf_param"] == {"alpha1": float(fx(name, "cbf_param/alpha1")), "alpha2": float(fx(name, "cbf_param/alpha2"))}
        assert np.array_equal(mc["u_lo"], fx(name, "u_lo")) and np.array_equal(mc["u_hi"], fx(name, "u_hi"))
        assert mc["nx"] == int(fx(name, "n_states"))
    for name in ORACLE_MODELS + ["VTOL2D"]:
        assert int(fx(name, "horizon")) == int(fx(name, "n_horizon_param")) == (30 if name == "VTOL2D" else 10)
        assert np.array_equal(fx(name, "rterm_u"), fx(name, "R"))          # set_rterm(u=R), mpc_cbf.py:180
        assert float(fx(name, "t_step")) == 0.05
        assert int(fx(name, "override/horizon")) == (30 if name == "VTOL2D" else 7)   # VTOL2D hard-codes 30 (mpc_cbf.py:41)


def test_oracle_tables_equal_the_reference_tables():
    """The constants typed into the oracles (DEFAULTS, model dicts) against the same reference tables."""
    du = M.DEFAULTS
    assert np.array_equal(np.diag(du["Q"]), fx("DynamicUnicycle2D", "Q")) and np.array_equal(du["R"], fx("DynamicUnicycle2D", "R"))
    assert (du["alpha1"], du["alpha2"]) == (float(fx("DynamicUnicycle2D", "cbf_param/alpha1")), float(fx("DynamicUnicycle2D", "cbf_param/alpha2")))
    assert du["N"] == int(fx("DynamicUnicycle2D", "horizon"))
    # state box: |v| <= v_max on state 3 only (mpc_cbf.py:193-196)
    lo, hi = fx("DynamicUnicycle2D", "x_lo"), fx("DynamicUnicycle2D", "x_hi")
    assert np.isinf(lo[:3]).all() and np.isinf(hi[:3]).all() and (lo[3], hi[3]) == (-du["v_max"], du["v_max"])
    assert np.array_equal(fx("DynamicUnicycle2D", "u_hi"), [du["a_max"], du["w_max"]])
    un = MU.DEFAULTS
    assert np.array_equal(np.diag(un["Q"]), fx("Unicycle2D", "Q")) and un["alpha"] == float(fx("Unicycle2D", "cbf_param/alpha"))
    assert np.isinf(fx("Unicycle2D", "x_lo")).all()                        # no state bounds for the kinematic unicycle
    assert np.array_equal(fx("Unicycle2D", "u_hi"), [1.0, un["w_max"]])
    for name, mdl in (("SingleIntegrator2D", L.si_model()), ("Quad3D", L.quad3d_model()), ("DoubleIntegrator2D", G.di_model()),
                      ("Quad2D", G.quad2d_model(dict(f_min=3.0))), ("KinematicBicycle2D", G.kb_model(dict(a_max=0.5, radius=0.5)))):
        assert np.array_equal(np.diag(mdl["Q"]), fx(name, "Q")), name
        assert np.array_equal(mdl["R"], fx(name, "R")), name
        assert close(mdl["u_lo"], fx(name, "u_lo")) and close(mdl["u_hi"], fx(name, "u_hi")), name
        for k in ("alpha", "alpha1", "alpha2"):
            if k in mdl:
                assert mdl[k] == float(fx(name, f"cbf_param/{k}")), (name, k)
        xlo, xhi = fx(name, "x_lo"), fx(name, "x_hi")
        boxed = {idx: (lo, hi) for idx, lo, hi in mdl.get("xb", [])}
        for i in range(mdl["nx"]):
            if i in boxed:
                assert close(boxed[i], (xlo[i], xhi[i])), (name, i)
            else:
                assert np.isinf(xlo[i]) and np.isinf(xhi[i]), (name, i)


def test_gain_overrides():
    """mpc_cbf_alpha / alpha1 / alpha2 in the robot spec replace the table values (mpc_cbf.py:90-95); mpc_horizon sets N (:15)."""
    from safe_control_amd.position_control.mpc_cbf import apply_mpc_overrides, default_mpc_cbf_param
    ov = dict(mpc_cbf_alpha=0.31, mpc_cbf_alpha1=0.27, mpc_cbf_alpha2=0.19)
    for name in ("DynamicUnicycle2D", "Unicycle2D"):
        got = apply_mpc_overrides(default_mpc_cbf_param(name), ov)
        keys = [str(k) for k in fx(name, "override/keys")]
        assert sorted(got) == keys
        assert [got[k] for k in keys] == [float(v) for v in fx(name, "override/vals")]


@pytest.mark.parametrize("name", ["DynamicUnicycle2D", "Quad3D", "VTOL2D"])
def test_update_tvp_padding(name):
    """update_tvp / tvp_fun (mpc_cbf.py:261-293, :338-364): None / empty -> dummies, 3-wide rows zero-padded, rows beyond
    num_obs dropped, 5-wide rows raise; goal padded with zeros to n_states."""
    from safe_control_amd.position_control.mpc_cbf import pad_obstacles as host_pad
    cases = {"none": None, "empty": [], "three_wide": [[1.0, 2.0, 0.3], [4.0, 5.0, 0.6]],
             "seven_wide": [[1.0, 2.0, 0.3, 0.1, -0.2, 0.0, 0.0], [3.0, 3.0, 0.5, 0.8, 4.0, 0.3, 1.0]],
             "too_many": [[float(j), float(j) + 1.0, 0.2 + 0.1 * j] for j in range(6)]}
    K = fx(name, "tvp/none/obs").shape[0]
    for cname, ob in cases.items():
        want = fx(name, f"tvp/{cname}/obs")
        assert np.array_equal(fx(name, f"tvp/{cname}/obs_attr"), want)
        for pad in (host_pad, M.pad_obstacles):
            assert np.array_equal(pad(None if ob is None else [np.array(o) for o in ob], K), want), (cname, pad.__module__)
        g = fx(name, f"tvp/{cname}/goal")
        nx = int(fx(name, "n_states"))
        ng = 3 if name == "Quad3D" else 2
        assert g.shape == (nx,) and np.array_equal(g[:ng], np.array([3.5, -1.25, 2.0])[:ng]) and not g[ng:].any()
    assert bool(fx(name, "tvp/five_wide_raises"))
    for pad in (host_pad, M.pad_obstacles):
        with pytest.raises(ValueError):
            pad([np.array([1.0, 2.0, 0.3, 0.0, 0.0])], K)


@pytest.mark.parametrize("name", ["KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"])
def test_full_state_barriers_equal_agent_barrier_dt_and_their_derivatives_are_exact(name):
    """h(x_k) and h(step(x_k, u_k)) - h(x_k) of oracle/mpc_kb_state.py against (h_k, d_h) of the reference's agent_barrier_dt as the MPC
    calls it (obstacle row 1 x 7: the velocity columns are never read); gradient and Hessian of the second-order forward mode against
    central differences."""
    mdl = (KS.c3bf_model if name.endswith("C3BF") else KS.dpcbf_model)(dict(spec_of(name), radius=float(fx(name, "robot_radius"))))
    P = KS.params(mdl, N=1)
    x, u, obs, hk, dh = fx(name, "x"), fx(name, "u"), fx(name, "obs"), fx(name, "hk"), fx(name, "dh")
    assert np.abs(obs[:, :, 3:5]).max() > 0.1                            # the draws DO carry obstacle velocities
    for i in range(x.shape[0]):
        y1 = G.kb_S(x[i], u[i], mdl["spec"], 0.05)
        for j in range(obs.shape[1]):
            h0 = KS.barrier(x[i], obs[i, j], P, False)[0]
            h1 = KS.barrier(y1, obs[i, j], P, False)[0]
            big = max(1.0, abs(h0))
            assert abs(h0 - hk[i, j]) <= 1e-12 * big and abs((h1 - h0) - dh[i, j]) <= 1e-12 * big, (i, j)
    rng = np.random.default_rng(0)
    for t in range(40):
        xv = np.array([rng.uniform(0, 5), rng.uniform(0, 5), rng.uniform(-3, 3), rng.uniform(0.3, 3)])
        ang = rng.uniform(-np.pi, np.pi); rho = rng.uniform(1.8, 4.0)
        ob = np.array([xv[0] + rho * np.cos(ang), xv[1] + rho * np.sin(ang), rng.uniform(0.2, 0.8), 0.3, -0.2, 0, 0])
        h, g, H = KS.barrier(xv, ob, P)
        eps = 1e-6
        for a in range(4):
            e = np.zeros(4); e[a] = eps
            hp, gp, _ = KS.barrier(xv + e, ob, P); hm, gm, _ = KS.barrier(xv - e, ob, P)
            assert abs((hp - hm) / (2 * eps) - g[a]) <= 1e-7 * max(1.0, np.abs(g).max())
            assert np.abs((gp - gm) / (2 * eps) - H[a]).max() <= 1e-6 * max(1.0, np.abs(H).max())
        assert np.abs(H - H.T).max() <= 1e-14 * max(1.0, np.abs(H).max())
