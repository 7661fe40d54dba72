"""CPU tests: the C restatement (oracle/c) against the numpy oracle and the golden vectors."""
import os

import numpy as np
import pytest

from oracle import c_oracle, cbf_qp, robots as R
from safe_control_amd import workloads as W


def spec_for(model):
    s = R.default_spec(model)
    if model == R.MODEL_DU:
        s.update(a_max=1.0, w_max=0.5, radius=0.25)
    else:
        s.update(a_max=5.0, radius=0.3)
    return s


GROUPS = {"du_circle": (R.MODEL_DU, "cbf"), "du_circle_hard": (R.MODEL_DU, "hard"),
          "du_superellipsoid": (R.MODEL_DU, "cbf"), "du_mixed_trunc": (R.MODEL_DU, "cbf"),
          "du_overlap": (R.MODEL_DU, "cbf"), "kb_circle": (R.MODEL_KB, "cbf"), "c3bf": (R.MODEL_KB_C3BF, "cbf"),
          "c3bf_k16": (R.MODEL_KB_C3BF, "cbf"), "dpcbf": (R.MODEL_KB_DPCBF, "cbf")}


@pytest.mark.parametrize("gname", list(GROUPS))
def test_c_oracle_on_golden_cases(golden_dir, gname):
    g = np.load(os.path.join(golden_dir, "cbfqp_cases.npz"))
    model, mode = GROUPS[gname]
    num_obs = int(g[f"{gname}/meta"][0])
    X, ur, obs, ks = g[f"{gname}/X"], g[f"{gname}/u_ref"], g[f"{gname}/obs"], g[f"{gname}/k"]
    n = len(X)
    o = np.zeros((n, num_obs, 7))
    kk = np.minimum(ks, num_obs).astype(np.int32)
    for i in range(n):
        o[i, : kk[i]] = obs[i, : kk[i]]
    u, st, h = c_oracle.cbfqp_batch(model, X, ur, o, spec_for(model), cbf_qp.default_cbf_param(model), 0.05, mode, kk)
    assert np.array_equal(st, g[f"{gname}/status_oracle"])
    ok = st == 0
    np.testing.assert_allclose(u[ok], g[f"{gname}/u_star_oracle"][ok], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("model,K", [(R.MODEL_DU, 8), (R.MODEL_KB, 5), (R.MODEL_KB_C3BF, 16), (R.MODEL_KB_DPCBF, 10)])
def test_c_oracle_equals_numpy_oracle(model, K):
    spec = spec_for(model)
    if model == R.MODEL_DU:
        X, goal, ur, obs = W.du_cbfqp_batch(300, K, seed=4)
    else:
        X, goal, ur, obs = W.kb_c3bf_batch(300, K, seed=4)
    cp = cbf_qp.default_cbf_param(model)
    u1, s1, h1 = cbf_qp.solve_batch(model, X, ur, obs, spec, cp)
    u2, s2, h2 = c_oracle.cbfqp_batch(model, X, ur, obs, spec, cp, n_threads=2)
    assert np.array_equal(s1, s2)
    ok = s1 == 0
    np.testing.assert_allclose(u1[ok], u2[ok], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(h1, h2, rtol=1e-10, atol=1e-10)


def test_workload_nominal_inputs_match_oracle():
    X, goal, ur, obs = W.du_cbfqp_batch(200, 8, seed=0)
    spec = spec_for(R.MODEL_DU)
    for i in range(200):
        np.testing.assert_allclose(R.nominal_input(R.MODEL_DU, X[i], goal[i], spec), ur[i], rtol=1e-12, atol=1e-12)
    X, goal, ur, obs = W.kb_c3bf_batch(200, 4, seed=0)
    spec = spec_for(R.MODEL_KB_C3BF)
    for i in range(200):
        np.testing.assert_allclose(R.nominal_input(R.MODEL_KB_C3BF, X[i], goal[i], spec), ur[i], rtol=1e-12, atol=1e-12)
    # every generated obstacle starts outside the inflated radius (h > 0)
    X, goal, ur, obs = W.du_cbfqp_batch(1000, 8, seed=0)
    d = np.hypot(obs[:, :, 0] - X[:, None, 0], obs[:, :, 1] - X[:, None, 1])
    assert np.all(d ** 2 - 1.01 * (obs[:, :, 2] + 0.25) ** 2 > 0)


def test_c_oracle_unicycle2d_on_reference_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "unicycle2d.npz"))
    G = {k.split("/", 1)[1]: g[k] for k in g.files}
    m = R.MODEL_UNI
    n = len(G["X"])
    kk = G["k"].astype(np.int32)
    o = np.zeros((n, 6, 7))
    for i in range(n):
        o[i, : kk[i]] = G["obs"][i][: kk[i]]
    u, st, h = c_oracle.cbfqp_batch(m, G["X"], G["u_ref"], o, R.default_spec(m), cbf_qp.default_cbf_param(m), 0.05, "cbf", kk)
    assert np.array_equal(st, G["status_oracle"])
    ok = st == 0
    np.testing.assert_allclose(u[ok], G["u_star_oracle"][ok], rtol=1e-8, atol=1e-8)


def _ms_one(a):
    from oracle import ms_ipopt as MS
    os.environ["OMP_NUM_THREADS"] = "1"
    u, st, it = MS.solve(MS.du_model(), a[0], a[1], a[2], a[3], opts=MS.KERNEL_PROFILE)
    return u, st, it


def test_compiled_cpu_baseline_of_config3_equals_the_numpy_oracle():
    """oracle/c/mpc_du_ms_cpu.cpp -- the multiple-shooting MPC-CBF solve compiled for the host cores (bench.py's cpu_baseline of BASELINE
    configs[2]; the 64 lanes of the kernel's solver header as fibers of one thread, OpenMP over problems) -- against oracle/ms_ipopt.py on the
    first 256 configs[2] problems: same status and iteration count on every one (the infeasible tenth with its restoration phase included),
    u0 to 1e-9; one thread and all threads give the same bits."""
    from multiprocessing import Pool
    n = 256
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("du", 4096, 8, seed=0))
    u, st, it = c_oracle.du_ms_cpu_batch(X, up, goal, obs, n_threads=0)
    u1, st1, it1 = c_oracle.du_ms_cpu_batch(X[:32], up[:32], goal[:32], obs[:32], n_threads=1)
    assert np.array_equal(u[:32], u1) and np.array_equal(st[:32], st1) and np.array_equal(it[:32], it1)
    with Pool(min(8, os.cpu_count() or 2)) as p:
        res = p.map(_ms_one, [(X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=4)
    so = np.array([r[1] for r in res]); ito = np.array([r[2] for r in res]); uo = np.array([r[0] for r in res])
    assert np.array_equal(st, so) and (so == 1).sum() >= 10
    assert (it != ito).sum() <= 2 and np.abs(it - ito).max() <= 1
    assert np.abs(u - uo)[it == ito].max() <= 1e-9


def _ms_one_di(a):
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import ms_ipopt as MS
    from safe_control_amd.robots.spec import complete_robot_spec
    sp = complete_robot_spec({"model": "DoubleIntegrator2D"})
    mdl = MS.di_model({k: v for k, v in sp.items() if k in MS.di_model()["spec"]})
    return MS.solve(mdl, a[0], a[1], a[2], a[3], opts=dict(MS.KERNEL_PROFILE))


def test_solver_header_for_double_integrator_equals_the_numpy_oracle():
    """The DoubleIntegrator2D instantiation of csrc/mpc_du_ms_solver.hpp (inputs held swapped, no state bound, robot.step's velocity rescaling
    and its curvature inside the barrier rows) compiled for the host, against oracle/ms_ipopt.py: di_model() on 192 bench draws and on 128
    draws that start above v_max with a random last input -- same status, same iteration count, u0 to 1e-9."""
    from multiprocessing import Pool
    n = 192
    X, up, goal, obs = (a[:n + 128].copy() for a in W.mpc_family_batch("di", 4096, 8, seed=0))
    rng = np.random.default_rng(5)
    X[n:, 2:4] = rng.uniform(-1.3, 1.3, (128, 2)); up[n:] = rng.uniform(-1.0, 1.0, (128, 2))
    u, st, it = c_oracle.du_ms_cpu_batch(X, up, goal, obs, n_threads=0, model="DoubleIntegrator2D")
    with Pool(min(8, os.cpu_count() or 2)) as p:
        res = p.map(_ms_one_di, [(X[i], up[i], goal[i], obs[i]) for i in range(n + 128)], chunksize=4)
    so = np.array([r[1] for r in res]); ito = np.array([r[2] for r in res]); uo = np.array([r[0] for r in res])
    assert np.array_equal(st, so) and (so[:n] == 1).sum() >= 8 and (so[n:] == 1).sum() >= 8
    assert (it != ito).sum() <= 2 and np.abs(it - ito).max() <= 1
    assert np.abs(u - uo)[it == ito].max() <= 1e-9


def _ms_one_kb(a):
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import ms_ipopt as MS
    from safe_control_amd.robots.spec import complete_robot_spec
    sp = complete_robot_spec({"model": "KinematicBicycle2D"})
    mdl = MS.kb_model({k: v for k, v in sp.items() if k in MS.kb_model()["spec"]})
    return MS.solve(mdl, a[0], a[1], a[2], a[3], opts=dict(MS.KERNEL_PROFILE, max_iter=150))


def test_solver_header_for_kinematic_bicycle_equals_the_numpy_oracle():
    """The KinematicBicycle2D instantiation of csrc/mpc_du_ms_solver.hpp (general stage layout: the inputs enter the positions; robot.step's speed
    clip inside the barrier rows; curvature of the bilinear dynamics) compiled for the host, against oracle/ms_ipopt.py: kb_model() on 160 bench
    draws at an iteration limit of 150: same status everywhere; same iteration count and u0 to 1e-8 on the solves that end within 60 iterations
    (the others cycle around the clip's kink, on both sides, along paths that rounding separates)."""
    from multiprocessing import Pool
    n = 160
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("kb", 4096, 8, seed=0))
    u, st, it = c_oracle.du_ms_cpu_batch(X, up, goal, obs, n_threads=0, model="KinematicBicycle2D", ipopt=dict(max_iter=150))
    with Pool(min(8, os.cpu_count() or 2)) as p:
        res = p.map(_ms_one_kb, [(X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=4)
    so = np.array([r[1] for r in res]); ito = np.array([r[2] for r in res]); uo = np.array([r[0] for r in res])
    assert np.array_equal(st, so)
    short = ito < 60
    assert short.mean() >= 0.9 and (it != ito)[short].sum() <= 2 and np.abs(it - ito)[short].max() <= 1
    assert np.abs(u - uo)[short & (it == ito)].max() <= 1e-8


def _ms_one_uni(a):
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import ms_ipopt as MS
    return MS.solve(MS.uni_model(), a[0], a[1], a[2], a[3], opts=dict(MS.KERNEL_PROFILE))


def test_solver_header_for_unicycle2d_equals_the_numpy_oracle():
    """The Unicycle2D instantiation of csrc/mpc_du_ms_solver.hpp (three states held as four with an idle one, inputs (v, omega) entering the
    positions, one-step barrier rows) compiled for the host, against oracle/ms_ipopt.py: uni_model() on 192 bench draws."""
    from multiprocessing import Pool
    n = 192
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("uni", 4096, 8, seed=0))
    u, st, it = c_oracle.du_ms_cpu_batch(X, up, goal, obs, n_threads=0, model="Unicycle2D")
    with Pool(min(8, os.cpu_count() or 2)) as p:
        res = p.map(_ms_one_uni, [(X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=4)
    so = np.array([r[1] for r in res]); ito = np.array([r[2] for r in res]); uo = np.array([r[0] for r in res])
    assert np.array_equal(st, so) and (it != ito).sum() <= 2 and np.abs(it - ito).max() <= 1
    assert np.abs(u - uo)[it == ito].max() <= 1e-9


def _ms_one_si(a):
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import ms_ipopt as MS
    return MS.solve(MS.si_model(), a[0], a[1], a[2], a[3], opts=dict(MS.KERNEL_PROFILE))


def test_solver_header_for_single_integrator_equals_the_numpy_oracle():
    """The SingleIntegrator2D instantiation of csrc/mpc_du_ms_solver.hpp (two states held as four with two idle ones, one-step rows, no curvature)
    compiled for the host, against oracle/ms_ipopt.py: si_model() on 160 bench draws."""
    from multiprocessing import Pool
    n = 160
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("si", 4096, 8, seed=0))
    u, st, it = c_oracle.du_ms_cpu_batch(X, up, goal, obs, n_threads=0, model="SingleIntegrator2D")
    with Pool(min(8, os.cpu_count() or 2)) as p:
        res = p.map(_ms_one_si, [(X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=4)
    so = np.array([r[1] for r in res]); ito = np.array([r[2] for r in res]); uo = np.array([r[0] for r in res])
    assert np.array_equal(st, so) and (it != ito).sum() <= 2 and np.abs(it - ito).max() <= 1
    assert np.abs(u - uo)[it == ito].max() <= 1e-9


def _ms_one_sweep(a):
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import ms_ipopt as MS
    from safe_control_amd.robots.spec import complete_robot_spec
    fam, N, x, up, g, ob = a
    mk = {"du": MS.du_model, "di": MS.di_model, "kb": MS.kb_model, "uni": MS.uni_model, "si": MS.si_model}[fam]
    sp = complete_robot_spec(dict({"model": W.MPC_FAMILIES[fam]}, **({"a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25} if fam == "du" else {})))
    return MS.solve(mk({k: v for k, v in sp.items() if k in mk()["spec"]}), x, up, g, ob, N=N, opts=dict(MS.KERNEL_PROFILE, max_iter=150))


@pytest.mark.parametrize("N,K", [(5, 1), (20, 3), (40, 16)])
def test_solver_header_over_horizons_and_obstacle_counts(N, K):
    """Every instantiation of csrc/mpc_du_ms_solver.hpp at horizons 5 / 20 / 40 (four, two and one lane per stage) with 1 / 3 / 16 obstacle
    slots, host build against the oracle: same status everywhere, same iteration count and u0 to 1e-8 on the solves that end within 60 iterations."""
    from multiprocessing import Pool
    n = 10
    for fam in ("du", "di", "uni", "si", "kb"):
        X, up, goal, obs = (a[:n].copy() for a in W.mpc_family_batch(fam, 64, K, seed=N))
        u, st, it = c_oracle.du_ms_cpu_batch(X, up, goal, obs, model=W.MPC_FAMILIES[fam], horizon=N, ipopt=dict(max_iter=150),
                                             spec=({"a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25} if fam == "du" else None))
        with Pool(min(8, os.cpu_count() or 2)) as p:
            res = p.map(_ms_one_sweep, [(fam, N, X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=2)
        so = np.array([r[1] for r in res]); ito = np.array([r[2] for r in res]); uo = np.array([r[0] for r in res])
        assert np.array_equal(st, so), (fam, N, K)
        short = ito < 60
        assert (it != ito)[short].sum() <= 1 and np.abs(u - uo)[short & (it == ito)].max() <= 1e-8, (fam, N, K)
