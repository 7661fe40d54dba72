"""oracle/ms_ipopt.py: the multiple-shooting NLP do-mpc hands to IPOPT, and the restatement of IPOPT's algorithm that solves it.

What pins what:
  * problem functions (dynamics rows, CBF rows, cost) on the reference's own code through tests/golden/mpc_functions.npz, like
    the condensed oracles (tests/test_oracle_mpc_golden.py);
  * derivatives (second-order forward mode over the stages) on central differences;
  * the solver on the one published IPOPT run that can be checked without IPOPT: problem HS071 of the IPOPT documentation
    (its tutorial problem) -- the starting line of the iteration log (objective 1.6109693e+01, inf_pr 1.12e+01, inf_du 5.28e-01:
    bound push, slack initialisation and least-square multipliers) and the published solution; beyond that the solver is
    **unpinned** (no IPOPT in the image), which is stated in its docstring;
  * the two formulations against each other: on config-3 draws the multiple-shooting solve from do-mpc's start and the condensed
    single-shooting oracle end in the same local optimum.
"""
import os

import os

import numpy as np
import pytest

from oracle import mpc_cbf as M
from oracle import ms_ipopt as MS

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mpc_functions.npz"))


def fx(name, key):
    return GOLD[f"{name}/{key}"]


def spec_of(name):
    return dict(zip([str(k) for k in fx(name, "spec_keys")], [float(v) for v in fx(name, "spec_vals")]))


def model_of(name):
    sp = spec_of(name)
    R = float(fx(name, "robot_radius"))
    if name == "VTOL2D":
        keys = MS.vtol_model()["spec"].keys()
        return MS.vtol_model(dict({k: v for k, v in sp.items() if k in keys}, radius=R))
    if name == "SingleIntegrator2D":
        return MS.si_model(dict(v_max=sp["v_max"], radius=R))
    if name == "Unicycle2D":
        return MS.uni_model(dict(v_max=sp["v_max"], w_max=sp["w_max"], radius=R))
    if name == "DoubleIntegrator2D":
        keys = MS.di_model()["spec"].keys()
        return MS.di_model(dict({k: v for k, v in sp.items() if k in keys}, radius=R))
    if name == "KinematicBicycle2D":
        keys = MS.kb_model()["spec"].keys()
        return MS.kb_model(dict({k: v for k, v in sp.items() if k in keys}, radius=R))
    if name in ("KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"):
        keys = MS.kb_model()["spec"].keys()
        return MS.kb_state_model(name, dict({k: v for k, v in sp.items() if k in keys}, radius=R))
    return MS.du_model(dict(v_max=sp["v_max"], a_max=sp["a_max"], w_max=sp["w_max"], radius=R))


@pytest.mark.parametrize("name", ["DynamicUnicycle2D", "VTOL2D", "KinematicBicycle2D", "DoubleIntegrator2D", "Unicycle2D", "SingleIntegrator2D", "KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"])
def test_rows_and_cost_of_a_one_stage_problem_equal_the_reference(name):
    """N = 1, w = [x, u, x_next]: the dynamics rows vanish at the reference's x_next (mpc_cbf.py:138-141), the inequality rows are the
    registered -cbf (:304), the objective is l(x) + m(x_next) (:144,176-178) + the rterm on u - u_prev (:180)."""
    mdl = model_of(name)
    x, u, goal, obs = fx(name, "x"), fx(name, "u"), fx(name, "goal"), fx(name, "obs")
    cons, xn, c0, c1, Rw = fx(name, "cons"), fx(name, "x_next"), fx(name, "cost"), fx(name, "cost_next"), fx(name, "rterm_u")
    assert np.array_equal(np.diag(fx(name, "Q")), mdl["Q"]) and np.array_equal(Rw, mdl["R"]) and int(fx(name, "horizon")) == mdl["N"]
    assert np.array_equal(fx(name, "u_lo"), mdl["u_lo"]) and np.array_equal(fx(name, "u_hi"), mdl["u_hi"])
    assert np.allclose(fx(name, "x_lo"), mdl["x_lo"], rtol=1e-15) and np.allclose(fx(name, "x_hi"), mdl["x_hi"], rtol=1e-15)
    for i in range(x.shape[0]):
        up = 0.3 * u[i]
        nlp = MS.StageNLP(mdl, x[i], up, goal[i], obs[i], N=1)
        w = np.concatenate([x[i], u[i], xn[i]])
        for level in (0, 2):
            ev = nlp.evaluate(w, level)
            assert np.abs(ev["c"]).max() <= 1e-12 * max(1.0, np.abs(xn[i]).max()), (name, i, level)
            big = max(1.0, np.abs(obs[i][:, 0:2]).max() ** 2)
            assert np.abs(ev["d"] - cons[i]).max() <= 1e-12 * max(1.0, np.abs(cons[i]).max()) + 1e-15 * big, (name, i, level)
            want = c0[i] + c1[i] + float(np.sum(Rw * (u[i] - up) ** 2))
            assert abs(ev["f"] - want) <= 1e-11 * max(1.0, abs(want)), (name, i, level)


@pytest.mark.parametrize("name", ["DynamicUnicycle2D", "VTOL2D", "KinematicBicycle2D", "DoubleIntegrator2D", "Unicycle2D", "SingleIntegrator2D", "KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"])
def test_derivatives_against_central_differences(name):
    mdl = model_of(name)
    rng = np.random.default_rng(3)
    x, u, goal, obs = fx(name, "x"), fx(name, "u"), fx(name, "goal"), fx(name, "obs")
    N = 3
    for i in (0, 7, 19):
        nlp = MS.StageNLP(mdl, x[i], 0.5 * u[i], goal[i], obs[i], N=N)
        w = nlp.initial_guess() + 0.05 * rng.standard_normal(nlp.n)
        w[nlp.iu] = u[i] + 0.02 * rng.standard_normal((N, mdl["nu"]))
        ev = nlp.evaluate(w, 2)
        yc, yd = rng.standard_normal(nlp.m_c), rng.standard_normal(nlp.m_d)
        W = ev["hess"](0.7, yc, yd)
        assert np.abs(W - W.T).max() <= 1e-12 * max(1.0, np.abs(W).max())
        eps = 1e-6
        hmax = np.abs(ev["d"]).max() / (mdl.get("alpha") or mdl["alpha1"] * mdl["alpha2"])      # ~ max |h|: the row is a difference of two or three such values
        gL = lambda e: 0.7 * e["grad"] + e["Jc"].T @ yc + e["Jd"].T @ yd
        for j in range(nlp.n):
            dw = np.zeros(nlp.n); dw[j] = eps
            ep, em = nlp.evaluate(w + dw, 2), nlp.evaluate(w - dw, 2)
            sc = max(1.0, abs(ev["grad"][j]))
            assert abs((ep["f"] - em["f"]) / (2 * eps) - ev["grad"][j]) <= 2e-6 * sc, (name, i, j)
            assert np.abs((ep["c"] - em["c"]) / (2 * eps) - ev["Jc"][:, j]).max() <= 1e-6 * max(1.0, np.abs(ev["Jc"][:, j]).max())
            # (a superellipsoid row of exponent 6 far from its obstacle has |d| ~ 4e4 and large third derivatives: truncation and round-off
            # of the difference quotient)
            assert np.abs((ep["d"] - em["d"]) / (2 * eps) - ev["Jd"][:, j]).max() <= 2e-5 * max(1.0, np.abs(ev["Jd"][:, j]).max()) + 4e-10 * hmax
            assert np.abs((gL(ep) - gL(em)) / (2 * eps) - W[:, j]).max() <= 1e-4 * max(1.0, np.abs(W[:, j]).max()), (name, i, j)


class HS071:
    """Problem 71 of the Hock-Schittkowski collection, the tutorial problem of the IPOPT documentation:
    min x1 x4 (x1 + x2 + x3) + x3  s.t.  x1 x2 x3 x4 >= 25,  |x|^2 = 40,  1 <= x <= 5,  start (1, 5, 5, 1)."""
    n, m_c, m_d = 4, 1, 1
    d_lo, d_hi = np.array([25.0]), np.array([np.inf])

    def x_bounds(self):
        return np.ones(4), 5.0 * np.ones(4)

    def evaluate(self, x, level=2):
        out = dict(f=x[0] * x[3] * (x[0] + x[1] + x[2]) + x[2], c=np.array([np.sum(x * x) - 40.0]), d=np.array([np.prod(x)]))
        if level >= 2:
            out["grad"] = np.array([x[3] * (2 * x[0] + x[1] + x[2]), x[0] * x[3], x[0] * x[3] + 1, x[0] * (x[0] + x[1] + x[2])])
            out["Jc"] = 2 * x[None, :]
            out["Jd"] = np.array([[x[1] * x[2] * x[3], x[0] * x[2] * x[3], x[0] * x[1] * x[3], x[0] * x[1] * x[2]]])

            def hess(sf, yc, yd):
                H = np.zeros((4, 4))
                H[0, 0] = 2 * x[3]; H[0, 1] = x[3]; H[0, 2] = x[3]; H[0, 3] = 2 * x[0] + x[1] + x[2]; H[1, 3] = x[0]; H[2, 3] = x[0]
                G = np.zeros((4, 4))
                G[0, 1] = x[2] * x[3]; G[0, 2] = x[1] * x[3]; G[0, 3] = x[1] * x[2]; G[1, 2] = x[0] * x[3]; G[1, 3] = x[0] * x[2]; G[2, 3] = x[0] * x[1]
                return sf * (H + np.triu(H, 1).T) + yc[0] * 2 * np.eye(4) + yd[0] * (G + G.T)
            out["hess"] = hess
        return out


def test_hs071_start_line_and_solution_of_the_ipopt_documentation():
    tr = []
    r = MS.solve_nlp(HS071(), np.array([1.0, 5.0, 5.0, 1.0]), trace=tr)
    # iteration 0 of the documented log:   0  1.6109693e+01 1.12e+01 5.28e-01  -1.0
    assert f"{tr[0]['f']:.7e}" == "1.6109693e+01" and f"{tr[0]['pinf']:.2e}" == "1.12e+01" and f"{tr[0]['dinf']:.2e}" == "5.28e-01"
    assert r["status"] == "optimal" and r["iters"] <= 12
    assert np.abs(r["x"] - np.array([1.0, 4.74299963, 3.82114998, 1.37940829])).max() <= 1e-6      # the published minimiser
    assert abs(r["f"] - 17.0140173) <= 1e-6


def test_infeasible_problem_ends_in_the_restoration_with_the_least_violation():
    """min x^2  s.t.  x >= 1 and x <= -1 (as two inequality rows): the restoration converges to the minimiser of the l1 violation."""
    class Inf:
        n, m_c, m_d = 1, 0, 2
        d_lo, d_hi = np.array([-np.inf, -np.inf]), np.zeros(2)

        def x_bounds(self):
            return np.array([-np.inf]), np.array([np.inf])

        def evaluate(self, x, level=2):
            out = dict(f=float(x[0] ** 2), c=np.zeros(0), d=np.array([1.0 - x[0], x[0] + 1.5]))
            if level >= 2:
                out.update(grad=2 * x, Jc=np.zeros((0, 1)), Jd=np.array([[-1.0], [1.0]]), hess=lambda sf, yc, yd: np.array([[2.0 * sf]]))
            return out
    r = MS.solve_nlp(Inf(), np.array([0.3]))
    assert r["status"] == "local_infeasibility" and r["code"] == 1
    assert -1.5 - 1e-6 <= r["x"][0] <= 1.0 + 1e-6                      # any point between the two half lines has the least violation (2.5)


def test_multiple_shooting_and_condensed_solves_agree_on_config3_draws():
    """BASELINE configs[2] draws: same local optimum from do-mpc's start (x_k = x0) with the filter method as from the
    rollout of u_prev with the condensed l1-merit method; the dynamics rows are closed and every CBF row holds."""
    from safe_control_amd import workloads as W
    X, goal, _, obs = W.du_cbfqp_batch(24, 8, seed=0)
    mdl = MS.du_model()
    n_same = 0
    for i in range(24):
        u, st, it, info = MS.solve(mdl, X[i], np.zeros(2), goal[i], obs[i], return_info=True)
        uo, so, _ = M.solve(X[i], np.zeros(2), goal[i], obs[i])
        if st == 0:
            assert np.abs(info["c"]).max() <= 1e-8 and info["d"].max() <= 1e-7
        if st == 0 and so == 0:
            n_same += int(np.abs(u - uo).max() <= 1e-5)
    assert n_same >= 20, n_same


def _both(i_args):
    import os as _os
    _os.environ["OMP_NUM_THREADS"] = "1"
    x, up, g, ob = i_args
    u, st, it = MS.solve(MS.du_model(), x, up, g, ob)
    uo, so, ito = M.solve(x, up, g, ob)
    return u, st, uo, so


def test_where_the_nlp_has_no_feasible_point_the_two_formulations_return_different_inputs():
    """The round-5 review's parity hole, measured: on BASELINE configs[2] draws the condensed single-shooting solve (oracle/mpc_cbf.py: what
    kernel 3 runs) and the multiple-shooting solve under IPOPT's algorithm (what the reference runs: position_control/mpc_cbf.py:162-174,384)
    agree on WHICH problems have no feasible point, and where both end optimal they hold the same input -- but on the infeasible ones the
    returned input (the restoration's last iterate, which the reference applies: status is hard-wired 'optimal', :10) differs: by more than
    1e-4 on most of them, by more than 1e-3 on a quarter (tools/exp_ms_vs_condensed.py on all 4096: 436 pairs, median 5.3e-4, 29 % > 1e-3,
    max 1.0).  That is why the multiple-shooting kernel (csrc/mpc_du_ms.hip, kernel 13) is the default for this robot since round 6."""
    from multiprocessing import Pool
    from safe_control_amd import workloads as W
    n = 160
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("du", 4096, 8, seed=0))
    with Pool(min(8, os.cpu_count() or 2)) as p:
        res = p.map(_both, [(X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=4)
    st = np.array([r[1] for r in res]); so = np.array([r[3] for r in res])
    du = np.array([np.abs(r[0] - r[2]).max() for r in res])
    assert (st == so).mean() >= 0.98
    both_opt, both_inf = (st == 0) & (so == 0), (st == 1) & (so == 1)
    assert both_opt.sum() >= 130 and np.mean(du[both_opt] <= 1e-5) >= 0.97
    assert both_inf.sum() >= 10 and np.median(du[both_inf]) >= 1e-4 and du[both_inf].max() >= 1e-3


def _vtol_problem(i=0):
    from safe_control_amd import workloads as W
    X, up, goal, obs = W.mpc_family_batch("vtol", 8, 8, seed=0)
    return MS.vtol_model(), X[i], up[i], goal[i], obs[i]


@pytest.mark.parametrize("phase", ["regular", "restoration", "restoration_ineq"])
def test_riccati_recursion_solves_the_same_system_as_the_dense_factorisation(phase):
    """The linear algebra of csrc/mpc_vtol_ms.hip (rows condensed into the stage blocks, Riccati recursion over (dx_k, du_{k-1}) with
    defects; in IPOPT's restoration the dynamics rows are soft: parallel sums) against LAPACK's symmetric indefinite factorisation of the
    whole primal-dual system, at a random interior iterate with random right-hand sides."""
    mdl, x0, up, goal, obs = _vtol_problem(1)
    nlp = MS.StageNLP(mdl, x0, up, goal, obs)
    o = dict(MS.OPTS, resto_elastic="ineq" if phase == "restoration_ineq" else "all")
    Pr = MS._Regular(nlp, nlp.initial_guess(), o)
    P = Pr if phase == "regular" else MS._Resto(Pr, nlp.initial_guess(), o)
    rng = np.random.default_rng(1)
    A = MS._Algo(P, o)
    x = A.push(nlp.initial_guess() + 0.01 * rng.standard_normal(nlp.n), P.x_L, P.x_U, 0.01, 0.01)
    ev = P.evaluate(x, 2, 0.1)
    y = 0.1 * rng.standard_normal(P.m)
    sig_x = np.where(np.isfinite(P.x_L) | np.isfinite(P.x_U), rng.uniform(0.1, 10, P.n), 0.0)
    sig_t = rng.uniform(0.01, 100, P.nt)
    W = ev["hess"](y) + (5.0 * np.eye(P.n) if phase == "regular" else 0.0)
    rx, rt, rg = rng.standard_normal(P.n), rng.standard_normal(P.nt), rng.standard_normal(P.m)
    sols = []
    for ls in (None, "riccati"):
        o["linear_solver"] = ls
        A.delta_w_last = 0.0
        sols.append(A.factor(W, ev["J"], sig_x, sig_t, 0.1)(rx, rt, rg))
        assert A.last_delta[0] == 0.0
    for a, b in zip(*sols):
        assert np.abs(a - b).max() <= 1e-8 * max(1.0, np.abs(a).max())


def test_optimal_decay_stage_elimination_against_the_dense_factorisation():
    """vtol_od_model in the recursion: the decay rates leave each stage first, the rows entering the Schur complement one at a time
    (_od_eliminate, what csrc/mpc_vtol_ms.hip does).  (a) Same step as LAPACK's factorisation of the whole system at a random interior
    iterate.  (b) With one row per stage at E = 1e12 (an active row at mu ~ 1e-9) the sequential form still gives the step to 1e-6 where
    the assembled block (od_elimination = 'none': rows condensed first, six-input stage) is off in the first digit -- the reason it exists."""
    mdl, x0, up, goal, obs = _vtol_problem(1)
    mdl = MS.vtol_od_model()
    nlp = MS.StageNLP(mdl, x0, up, goal, obs)
    o = dict(MS.OPTS)
    P = MS._Regular(nlp, nlp.initial_guess(), o)
    rng = np.random.default_rng(2)
    A = MS._Algo(P, o)
    x = A.push(nlp.initial_guess() + 0.01 * rng.standard_normal(nlp.n), P.x_L, P.x_U, 0.01, 0.01)
    ev = P.evaluate(x, 2, 0.1)
    y = 1e-3 * rng.standard_normal(P.m)                                # (the decay block carries a1 a2 h0 y: keep it below the 5 I)
    sig_x = np.where(np.isfinite(P.x_L) | np.isfinite(P.x_U), rng.uniform(0.1, 10, P.n), 0.0)
    W = ev["hess"](y) + 5.0 * np.eye(P.n)
    rx, rt, rg = rng.standard_normal(P.n), rng.standard_normal(P.nt), rng.standard_normal(P.m)
    for stiff in (False, True):
        sig_t = rng.uniform(0.01, 100, P.nt)
        if stiff:
            sig_t[::8] = 1e12
        sols = {}
        for name, extra in (("dense", dict(linear_solver=None)), ("seq", dict(linear_solver="riccati")), ("assembled", dict(linear_solver="riccati", od_elimination="none"))):
            A.o = dict(o, **extra)
            A.delta_w_last = 0.0
            sols[name] = A.factor(W, ev["J"], sig_x, sig_t, 0.1)(rx, rt, rg)
            assert A.last_delta[0] == 0.0
        # (the multipliers of the stiff rows come back as E (a . dx - b): the rounding of dx times 1e12 -- they are left out of the stiff comparison;
        # the next iterate's residual is evaluated exactly, so the algorithm corrects them, which is what the GPU parity test sees)
        soft = np.ones(P.m, dtype=bool)
        if stiff:
            soft[P.t_row[::8]] = False
        pick = lambda sol: (sol[0], sol[1], sol[2][soft])
        err = {k: max(np.abs(a - b).max() / max(1.0, np.abs(a).max()) for a, b in zip(pick(sols[k]), pick(sols["dense"]))) for k in ("seq", "assembled")}
        assert err["seq"] <= (1e-6 if stiff else 1e-8), err
        if stiff:
            assert err["assembled"] >= 1e-3, err
        else:
            assert err["assembled"] <= 1e-8, err


def test_kernel_profile_follows_the_default_solve_and_hands_back_restorations():
    """linear_solver = riccati, max_soc = 0, restoration = none (what the HIP kernel runs): same iterates as the dense solve with second-order
    corrections on a feasible problem; on the first NLP of the reference's example scene (no feasible point) it stops with 'needs_resto' where
    the full algorithm enters its restoration phase and reports local infeasibility."""
    mdl, x0, up, goal, obs = _vtol_problem(0)
    prof = dict(linear_solver="riccati", max_soc=0, restoration="none")
    u1, s1, i1 = MS.solve(mdl, x0, up, goal, obs)
    u2, s2, i2 = MS.solve(mdl, x0, up, goal, obs, opts=prof)
    assert (s1, i1) == (s2, i2) == (0, i1) and np.abs(u1 - u2).max() <= 1e-10
    ob = np.hstack([np.array([[67.0, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[73.0, float(z), 0.5] for z in range(1, 7)]), np.zeros((10, 4))])
    mdl2 = MS.vtol_model(dict(radius=0.6, v_max=20.0))
    u, st, it, info = MS.solve(mdl2, np.array([2.0, 10.0, 0.0, 20.0, 0.0, 0.0]), np.zeros(4), [70.0, 10.0], ob, return_info=True, opts=prof)
    assert info["status"] == "needs_resto" and st == 4 and it < 100


def test_kernel_profile_restoration_reaches_the_same_verdict_as_the_full_algorithm():
    """KERNEL_PROFILE (what csrc/mpc_vtol_ms.hip runs since its restoration phase went into the kernel: elastic variables on the CBF rows only,
    dynamics rows hard inside the restoration, stall rule) on the first NLP of the reference's example scene: converged to a point of local
    infeasibility like the full algorithm (elastic variables on every row: a different restoration problem, hence another stationary point
    of the violation -- front thrust 1.0 against 0.73); the iteration count is the one the GPU test holds the kernel to."""
    ob = np.hstack([np.array([[67.0, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[73.0, float(z), 0.5] for z in range(1, 7)]), np.zeros((10, 4))])
    mdl = MS.vtol_model(dict(radius=0.6, v_max=20.0))
    x0 = np.array([2.0, 10.0, 0.0, 20.0, 0.0, 0.0])
    tr = []
    u1, s1, i1, info1 = MS.solve(mdl, x0, np.zeros(4), [70.0, 10.0], ob, return_info=True, opts=dict(MS.KERNEL_PROFILE), trace=tr)
    u2, s2, i2, info2 = MS.solve(mdl, x0, np.zeros(4), [70.0, 10.0], ob, return_info=True)
    assert info1["status"] == info2["status"] == "local_infeasibility" and s1 == s2 == 1
    assert i1 == 134 and sum(1 for q in tr if q["resto"]) >= 50
    assert u1[0] >= 0.7 and u2[0] >= 0.7 and u1[3] > 0.3 and u2[3] > 0.3   # (both: front thrust up, elevator up)


def _profile_and_full(i_args):
    import os as _os
    _os.environ["OMP_NUM_THREADS"] = "1"
    x, up, g, ob = i_args
    return MS.solve(MS.du_model(), x, up, g, ob, opts=dict(MS.KERNEL_PROFILE)), MS.solve(MS.du_model(), x, up, g, ob)


def test_kernel_profile_returns_the_full_algorithms_input_on_config3_draws():
    """What kernel 13 leaves out of IPOPT's algorithm (second-order corrections, elastic variables on the dynamics rows inside the
    restoration, the dense factorisation) does not move what BASELINE configs[2] returns: on the first 96 draws the kernel profile and the
    full restatement agree in status on every draw and in the applied input to 1e-8, the draws without a feasible point included (on the
    first 1024: one exception, two restorations ending in different local minimisers of the violation)."""
    from multiprocessing import Pool
    from safe_control_amd import workloads as W
    n = 96
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("du", 4096, 8, seed=0))
    with Pool(min(8, os.cpu_count() or 2)) as p:
        res = p.map(_profile_and_full, [(X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=4)
    sa = np.array([r[0][1] for r in res]); sb = np.array([r[1][1] for r in res])
    du = np.array([np.abs(r[0][0] - r[1][0]).max() for r in res])
    assert (sa == sb).all() and (sb == 1).sum() >= 8
    assert du.max() <= 1e-8
