"""A recording stand-in for the handful of cvxpy names position_control/backup_cbf_qp.py uses (:690-735), for
tests/golden/make_golden_backup.py ONLY (build container; cvxpy / OSQP are not installable here).

It lets the reference's own QP-assembly code run verbatim: ``Variable``, affine expressions built with ``@``, ``-``,
``>=``, ``<=``, ``sum_squares``, ``Minimize``, ``Problem``.  ``Problem.solve`` hands the recorded data

    minimise || W (x - x_ref) ||^2   s.t.   A x >= b  (all recorded rows, the box included)

to the oracle's exact active-set solver (oracle/qp.py) instead of OSQP: the minimiser of a strictly convex QP is unique,
so this pins what OSQP approximates (to its 1e-3-class ADMM tolerances).  The last problem solved is kept in ``LAST``."""
import numpy as np

LAST = {}
OSQP, SCS, GUROBI = "OSQP", "SCS", "GUROBI"


class _Expr:
    __array_ufunc__ = None                    # numpy defers `ndarray @ expr`, `ndarray - expr` to the reflected operators

    def __init__(self, var, A, b):
        self.var, self.A, self.b = var, np.atleast_2d(np.asarray(A, dtype=float)), np.asarray(b, dtype=float).reshape(-1)

    def __sub__(self, other):
        return _Expr(self.var, self.A, self.b - np.asarray(other, dtype=float).reshape(-1))

    def __add__(self, other):
        return _Expr(self.var, self.A, self.b + np.asarray(other, dtype=float).reshape(-1))

    def __rmatmul__(self, M):
        M = np.atleast_2d(np.asarray(M, dtype=float))
        return _Expr(self.var, M @ self.A, M @ self.b)

    def __ge__(self, rhs):                    # A x + b >= rhs
        return _Constraint(self.var, self.A, np.broadcast_to(np.asarray(rhs, dtype=float), self.b.shape) - self.b)

    def __le__(self, rhs):                    # A x + b <= rhs   ->   -A x >= b - rhs
        return _Constraint(self.var, -self.A, self.b - np.broadcast_to(np.asarray(rhs, dtype=float), self.b.shape))


class Variable(_Expr):
    def __init__(self, n):
        n = int(n)
        super().__init__(self, np.eye(n), np.zeros(n))
        self.n, self.value = n, None


class _Constraint:
    def __init__(self, var, A, b):            # A x >= b
        self.var, self.A, self.b = var, A, b


class _SumSquares:
    def __init__(self, e):
        self.e = e


def sum_squares(e):
    return _SumSquares(e)


class Minimize:
    def __init__(self, obj):
        self.obj = obj


class Problem:
    def __init__(self, objective, constraints):
        self.objective, self.constraints, self.status = objective, constraints, None

    def solve(self, **_):
        from oracle import qp as oqp
        e = self.objective.obj.e              # || W x - W x_ref ||^2 with W diagonal
        W = e.A
        assert np.allclose(W, np.diag(np.diag(W))) and np.all(np.diag(W) > 0)
        w = np.diag(W)
        x_ref = -e.b / w
        A = np.vstack([c.A for c in self.constraints])
        b = np.concatenate([c.b for c in self.constraints])
        # v = W x:  minimise ||v - v_ref||^2  s.t.  (A W^-1) v - b >= 0
        v, st = oqp.solve_qpn(A / w[None, :], -b, w * x_ref)
        LAST.clear()
        LAST.update(A=A.copy(), b=b.copy(), w=w.copy(), x_ref=x_ref.copy(), status=st)
        if st == 0:
            e.var.value = v / w
            self.status = "optimal"
        else:
            e.var.value = None
            self.status = "infeasible"
        LAST["x"] = None if e.var.value is None else e.var.value.copy()
        return None
