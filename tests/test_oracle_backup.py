"""CPU: oracle/backup_cbf.py (Backup-CBF QP) against tests/golden/backup_cbf.npz -- the reference's own
rollout, finite-difference sensitivities, rows and QP statement executed on its evade scenario
(tests/golden/make_golden_backup.py).  The QP minimiser in the fixture comes from the exact solver (OSQP absent)."""
import os

import numpy as np
import pytest

from oracle import backup_cbf as B

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "backup_cbf.npz"))


def test_environment_and_robot_constants():
    e, s = B.default_env(), B.default_spec()
    assert np.array_equal(G["env"], [e["hallway_length"], e["half_width"], e["pocket_x_min"], e["pocket_x_max"], e["pocket_y_min"],
                                     e["pocket_y_max"], e["goal_x_min"], e["goal_x_max"], e["bullet_speed"], e["bullet_length"],
                                     e["bullet_width"], e["bullet_start_x"]])
    assert np.array_equal(G["spec"], [s["radius"], s["a_max"], s["v_max"], s["safety_margin"], s["alpha"], s["alpha_terminal"]])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_single_calls_reproduce_the_reference(tag):
    dt, hor = float(G[f"{tag}_dt"]), float(G[f"{tag}_horizon"])
    X, BX = G[f"{tag}_X"], G[f"{tag}_bullet_x"]
    seen = set()
    for i in range(len(X)):
        u, info = B.solve(X[i], G[f"{tag}_u_nom"][i], BX[i], dt, hor, return_info=True)
        n = int(G[f"{tag}_n_rows"][i])
        assert np.abs(info["phi"] - G[f"{tag}_phi"][i]).max() <= 1e-12
        assert np.abs(info["S"] - G[f"{tag}_S"][i]).max() <= 1e-9          # forward differences: 1e-5 / eps amplification
        assert info["n_rows"] == n
        ref = G[f"{tag}_rows"][i][:n]
        assert np.abs(info["rows"] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
        assert info["qp_status"] == int(G[f"{tag}_qp_status"][i])
        assert info["using_backup"] == bool(G[f"{tag}_using_backup"][i])
        assert abs(info["h_min"] - float(G[f"{tag}_h_min"][i])) <= 1e-12
        assert np.abs(u - G[f"{tag}_u"][i]).max() <= 1e-9
        seen.add((info["qp_status"], info["using_backup"]))
    assert len(seen) >= 3                    # solved / infeasible with both fallbacks all occur among the cases


def test_closed_loop_reproduces_the_example():
    T = 260                                  # first bullet pass, the retreat into the pocket and the respawn
    X, U, Bx, UB, HM, outcome, _ = B.closed_loop(G["loop_X"][0], G["loop_bullet_x"][0], T)
    assert outcome == 0 and len(X) == T
    assert np.abs(X - G["loop_X"][:T]).max() <= 1e-8
    assert np.abs(U - G["loop_U"][:T]).max() <= 1e-6
    assert np.array_equal(UB, G["loop_using_backup"][:T])
    assert np.abs(Bx - G["loop_bullet_x"][:T]).max() <= 1e-12
    assert np.abs(HM - G["loop_h_min"][:T]).max() <= 1e-9
    assert X[:, 1].max() > 3.0               # the robot did retreat into the pocket


def test_rows_are_the_definition():
    """A kept row is grad h . S_i . (f0 + g0 u) - grad h . f_pi + dh/dt + alpha h >= 0 (backup_cbf_qp.py:650-661)."""
    i = 3
    x0, bx = G["a_X"][i], float(G["a_bullet_x"][i])
    env, spec = B.default_env(), B.default_spec()
    phi, S = B.rollout(x0, 120, 0.1, env, spec)
    lhs, rhs, keep, _ = B.assemble_rows(x0, phi, S, bx, 0.1, 12.0, env, spec)
    k = 40
    t = k * 0.1
    g = B._fd_grad(lambda z: B.h_safety(z, t, bx, env, spec), phi[k])
    u = np.array([0.3, -0.2])
    xdot = np.array([x0[2], x0[3], u[0], u[1]])
    h = B.h_safety(phi[k], t, bx, env, spec)
    dhdt = (B.h_safety(phi[k], t + 0.1, bx, env, spec) - h) / 0.1
    lhs_def = g @ S[k] @ xdot - g @ ((phi[k + 1] - phi[k]) / 0.1) + dhdt + spec["alpha"] * h
    assert abs((lhs[k - 1] @ u - rhs[k - 1]) - lhs_def) <= 1e-9
