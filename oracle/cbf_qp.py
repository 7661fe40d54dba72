"""Float64 restatement of CBFQP.solve_control_problem (position_control/cbf_qp.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np

from . import robots as R
from .qp import STATUS_INFEASIBLE, STATUS_OPTIMAL, box_rows, solve_qp2


def default_cbf_param(model):
    """CBF gains per model, position_control/cbf_qp.py:12-35."""
    if model in R.REL_DEG2:
        return {"alpha1": 1.5, "alpha2": 1.5}
    if model in (R.MODEL_SI, R.MODEL_UNI):                 # cbf_qp.py:12-15
        return {"alpha": 1.0}
    return {"alpha": 1.5}


def input_bounds(model, spec):
    """Input box, position_control/cbf_qp.py:62-65 (DU) and :70-73 (KB family)."""
    if model == R.MODEL_DU:
        hi = np.array([spec["a_max"], spec["w_max"]], dtype=np.float64)
    elif model == R.MODEL_SI:                              # cbf_qp.py:54-57
        hi = np.array([spec["v_max"], spec["v_max"]], dtype=np.float64)
    elif model == R.MODEL_UNI:                             # cbf_qp.py:58-61
        hi = np.array([spec["v_max"], spec["w_max"]], dtype=np.float64)
    elif model == R.MODEL_DI:                              # cbf_qp.py:66-69
        hi = np.array([spec["a_max"], spec["a_max"]], dtype=np.float64)
    elif model == R.MODEL_QUAD2D:                          # cbf_qp.py:74-79: f_min <= u <= f_max (not symmetric)
        return (np.array([spec["f_min"], spec["f_min"]], dtype=np.float64),
                np.array([spec["f_max"], spec["f_max"]], dtype=np.float64))
    else:
        hi = np.array([spec["a_max"], spec["beta_max"]], dtype=np.float64)
    return -hi, hi


def assemble_rows(model, X, obs_list, spec, cbf_param, num_obs, dt=0.05, cbf_mode="cbf"):
    """CBF rows ``A1 u + b1 >= 0`` and barrier values for one agent.

    position_control/cbf_qp.py:108-183: rows are zero-initialised (:110-111),
    one row per obstacle up to ``num_obs`` (:126-128), rel-deg-2 models use
    ``A = dh_dot_dx g``, ``b = dh_dot_dx f + (a1+a2) h_dot + a1 a2 h``
    (:167-183), rel-deg-1 models ``A = dh_dx g``, ``b = dh_dx f + alpha h``
    (:155-165); 'hard' mode variants :158-161 / :170-177.

    Returns ``A (num_obs,2), b (num_obs,), h (num_obs,)`` (h is NaN for unused rows).
    """
    X = np.asarray(X, dtype=np.float64).reshape(-1)
    A = np.zeros((num_obs, 2))
    b = np.zeros(num_obs)
    hv = np.full(num_obs, np.nan)
    fx = R.f(model, X, spec)
    gx = R.g(model, X, spec)
    radius = spec["radius"]
    row = 0
    for obs in obs_list:
        if obs is None:
            continue
        if row >= num_obs:
            break
        obs = np.asarray(obs, dtype=np.float64)
        if model in R.REL_DEG2:
            h, h_dot, dhd = R.agent_barrier(model, X, obs, radius)
            A[row] = dhd @ gx
            if cbf_mode == "hard":
                b[row] = h / dt ** 2 + 2.0 * h_dot / dt + dhd @ fx
            else:
                a1, a2 = cbf_param["alpha1"], cbf_param["alpha2"]
                b[row] = dhd @ fx + (a1 + a2) * h_dot + (a1 * a2) * h
        else:
            h, dh = R.agent_barrier(model, X, obs, radius)
            A[row] = dh @ gx
            if cbf_mode == "hard":
                b[row] = h / dt + dh @ fx
            else:
                b[row] = dh @ fx + cbf_param["alpha"] * h
        hv[row] = h
        row += 1
    return A, b, hv


def solve(model, X, u_ref, obs_list, spec, cbf_param=None, num_obs=10, dt=0.05, cbf_mode="cbf"):
    """One CBF-QP solve.  Returns dict(u, status, h, A, b).

    ``obs_list is None`` returns ``u_ref`` unclipped with status optimal
    (cbf_qp.py:113-118).  Infeasible -> ``u`` is None, status 1.
    """
    u_ref = np.asarray(u_ref, dtype=np.float64).reshape(2)
    if cbf_param is None:
        cbf_param = default_cbf_param(model)
    if obs_list is None:
        return dict(u=u_ref.copy(), status=STATUS_OPTIMAL, h=np.full(num_obs, np.nan),
                    A=np.zeros((num_obs, 2)), b=np.zeros(num_obs))
    A, b, hv = assemble_rows(model, X, obs_list, spec, cbf_param, num_obs, dt, cbf_mode)
    lo, hi = input_bounds(model, spec)
    Gb, cb = box_rows(lo, hi)
    u, status = solve_qp2(np.vstack([A, Gb]), np.concatenate([b, cb]), u_ref)
    return dict(u=u, status=status, h=hv, A=A, b=b)


def solve_batch(model, X, u_ref, obs, spec, cbf_param=None, dt=0.05, cbf_mode="cbf"):
    """Per-agent loop over a batch.  ``obs`` is (B,K,7) or shared (K,7).

    Returns ``u (B,2)`` (NaN where infeasible), ``status (B,) int32``, ``h (B,K)``.
    """
    X = np.asarray(X, dtype=np.float64)
    u_ref = np.asarray(u_ref, dtype=np.float64)
    obs = np.asarray(obs, dtype=np.float64)
    B = X.shape[0]
    K = obs.shape[-2]
    u = np.full((B, 2), np.nan)
    st = np.zeros(B, dtype=np.int32)
    h = np.zeros((B, K))
    for i in range(B):
        o = obs if obs.ndim == 2 else obs[i]
        r = solve(model, X[i], u_ref[i], list(o), spec, cbf_param, num_obs=K, dt=dt, cbf_mode=cbf_mode)
        st[i] = r["status"]
        if r["u"] is not None:
            u[i] = r["u"]
        h[i] = r["h"]
    return u, st, h
