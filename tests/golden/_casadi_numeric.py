"""Numeric stand-ins for the ``casadi`` / ``do_mpc`` names the reference's MPC path uses.

Used ONLY by tests/golden/make_golden.py in the build container, to EXECUTE the reference's own
``agent_barrier_dt``, ``f_casadi`` / ``g_casadi`` and ``MPCCBF.__init__ / create_model / create_mpc /
set_cbf_constraint / update_tvp`` (position_control/mpc_cbf.py, robots/*.py under /root/reference) on concrete
numbers.  casadi is a symbolic library; everything the reference does with it on this path is build an
expression from ~25 elementary functions, so evaluating those functions eagerly on float64 arrays yields the
value of the expression the reference would hand to IPOPT, at the point the "symbols" are bound to.

* ``SX`` is a small dense float64 matrix with casadi's conventions (always 2-D, ``x[i]`` is linear
  column-major indexing, ``x[i, j]`` returns a 1x1 matrix, ``*`` is element-wise, ``@`` / ``mtimes`` is the
  matrix product).  It is deliberately NOT an ndarray subclass: the reference's ``angle_normalize`` tests
  ``isinstance(x, np.ndarray)`` first and must take its casadi (``fmod``) branch for these values.
* ``do_mpc`` is a recorder: ``Model.set_variable`` binds a variable to the numbers in ``POINT``, and
  ``set_rhs / set_expression / set_objective / set_rterm / bounds[...] / set_nl_cons / set_param`` store what the
  reference passes.  No optimisation happens here (IPOPT is not available); the fixtures pin the PROBLEM.

Nothing in this file restates the reference, and nothing of the reference is copied into it.
"""
import math
import sys
import types

import numpy as np

POINT = {}          # variable name -> ndarray the next Model.set_variable(name) binds to


def _val(a):
    if isinstance(a, SX):
        return a.v
    arr = np.asarray(a)
    if arr.dtype == object:                # e.g. np.array([[cos(SX), 0], [sin(SX), 0], [0, 1]]) in the reference
        arr = np.vectorize(float, otypes=[np.float64])(arr)
    arr = arr.astype(np.float64, copy=False)
    if arr.ndim == 0:
        return arr.reshape(1, 1)
    if arr.ndim == 1:
        return arr.reshape(-1, 1)          # casadi turns a flat list into a column
    return arr


def _bc(a, b):
    """casadi broadcasts only 1x1 against a matrix."""
    a, b = _val(a), _val(b)
    if a.shape != b.shape and a.size != 1 and b.size != 1:
        raise ValueError(f"shape mismatch {a.shape} vs {b.shape}")
    return a, b


def _as_slice(i):
    """An integer index keeps its axis (casadi returns 1x1 / 1xn / nx1 matrices, never scalars)."""
    if isinstance(i, (int, np.integer)):
        return slice(i, i + 1 if i != -1 else None)
    return i


class SX:
    __array_priority__ = 1000

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        """np.cos(SX), ndarray @ SX, ndarray + SX ... evaluate on the values and stay SX (casadi overloads these too)."""
        if method != "__call__" or kw:
            return NotImplemented
        with np.errstate(all="ignore"):
            return SX(ufunc(*[_val(i) for i in inputs]))

    def __init__(self, v=0.0):
        self.v = np.array(_val(v), dtype=np.float64)

    # -- constructors casadi offers on the class
    @classmethod
    def zeros(cls, r, c=1):
        return cls(np.zeros((r, c)))

    @classmethod
    def eye(cls, n):
        return cls(np.eye(n))

    @classmethod
    def sym(cls, name, r=1, c=1):
        return cls(np.array(POINT.get(name, np.zeros((r, c))), dtype=np.float64).reshape(r, c))

    # -- shape
    @property
    def shape(self):
        return self.v.shape

    @property
    def T(self):
        return SX(self.v.T)

    def size1(self):
        return self.v.shape[0]

    def size2(self):
        return self.v.shape[1]

    def __float__(self):
        if self.v.size != 1:
            raise TypeError("only 1x1 converts to float")
        return float(self.v.reshape(-1)[0])

    def __bool__(self):
        return bool(float(self))

    # -- indexing
    def _linear(self, i):
        flat = self.v.reshape(-1, order="F")
        return flat[i]

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            r, c = (_as_slice(i) for i in idx)
            return SX(self.v[r, c])
        out = self._linear(idx)
        return SX(np.asarray(out).reshape(-1, 1))

    def __setitem__(self, idx, val):
        val = _val(val)
        if isinstance(idx, tuple):
            self.v[idx] = val.reshape(np.shape(self.v[idx])) if np.ndim(self.v[idx]) else float(val.reshape(-1)[0])
            return
        flat = self.v.reshape(-1, order="F")
        flat[idx] = val.reshape(-1)[0] if np.ndim(flat[idx]) == 0 else val.reshape(-1)
        self.v = flat.reshape(self.v.shape, order="F")

    # -- arithmetic (element-wise, 1x1 broadcasts)
    def _bin(self, other, fn, swap=False):
        a, b = _bc(self, other)
        with np.errstate(all="ignore"):          # if_else evaluates both branches; the unused one may overflow
            return SX(fn(b, a) if swap else fn(a, b))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __pow__(self, o): return self._bin(o, _pow)
    def __rpow__(self, o): return self._bin(o, _pow, True)
    def __neg__(self): return SX(-self.v)
    def __pos__(self): return self
    def __abs__(self): return SX(np.abs(self.v))
    def __matmul__(self, o): return SX(self.v @ _val(o))
    def __rmatmul__(self, o): return SX(_val(o) @ self.v)
    def __lt__(self, o): return self._bin(o, lambda a, b: (a < b).astype(np.float64))
    def __le__(self, o): return self._bin(o, lambda a, b: (a <= b).astype(np.float64))
    def __gt__(self, o): return self._bin(o, lambda a, b: (a > b).astype(np.float64))
    def __ge__(self, o): return self._bin(o, lambda a, b: (a >= b).astype(np.float64))

    def __repr__(self):
        return f"SX({self.v!r})"


class MX(SX):
    pass


class DM(SX):
    pass


def _pow(a, b):
    """casadi's pow on reals: integer-valued exponents keep the sign rules of repeated multiplication."""
    with np.errstate(all="ignore"):
        return np.power(a, b)


def _un(fn):
    def f(x):
        with np.errstate(all="ignore"):
            return SX(fn(_val(x)))
    return f


def _bi(fn):
    def f(a, b):
        a, b = _bc(a, b)
        with np.errstate(all="ignore"):
            return SX(fn(a, b))
    return f


def vertcat(*args):
    if not args:
        return SX(np.zeros((0, 1)))
    return SX(np.vstack([_val(a) for a in args]))


def horzcat(*args):
    if not args:
        return SX(np.zeros((1, 0)))
    return SX(np.hstack([_val(a) for a in args]))


def mtimes(*args):
    seq = args[0] if len(args) == 1 and isinstance(args[0], (list, tuple)) else args
    out = _val(seq[0])
    for m in seq[1:]:
        m = _val(m)
        out = out * m if (out.size == 1 or m.size == 1) else out @ m
    return SX(out)


def if_else(c, a, b):
    c = _val(c)
    a, b = _val(a), _val(b)
    return SX(np.where(c != 0, a, b))


def install_casadi():
    m = types.ModuleType("casadi")
    m.__dict__.update(
        SX=SX, MX=MX, DM=DM, pi=math.pi, inf=math.inf,
        cos=_un(np.cos), sin=_un(np.sin), tan=_un(np.tan), atan=_un(np.arctan), exp=_un(np.exp), log=_un(np.log),
        tanh=_un(np.tanh), sqrt=_un(np.sqrt), fabs=_un(np.abs), sign=_un(np.sign),
        atan2=_bi(np.arctan2), fmod=_bi(np.fmod), fmax=_bi(np.maximum), fmin=_bi(np.minimum), power=_bi(_pow),
        hypot=_bi(np.hypot),
        norm_2=lambda x: SX(np.linalg.norm(_val(x).reshape(-1))), sumsqr=lambda x: SX(np.sum(_val(x) ** 2)),
        vertcat=vertcat, horzcat=horzcat, mtimes=mtimes, if_else=if_else,
    )
    sys.modules["casadi"] = m
    return m


# ---------------------------------------------------------------------------------------------------------------
# do_mpc recorder
# ---------------------------------------------------------------------------------------------------------------
class _Struct(dict):
    """model.x['x'] / model.tvp['obs'] / model.aux['cost'] access."""


class Model:
    def __init__(self, kind):
        self.kind = kind
        self.x, self.u, self.tvp, self.aux = _Struct(), _Struct(), _Struct(), _Struct()
        self.shapes = {}
        self.rhs = {}

    def set_variable(self, var_type, var_name, shape=(1, 1)):
        if isinstance(shape, int):
            shape = (shape, 1)
        val = np.array(POINT.get(var_name, np.zeros(shape)), dtype=np.float64).reshape(shape)
        sx = SX(val)
        {"_x": self.x, "_u": self.u, "_tvp": self.tvp}[var_type][var_name] = sx
        self.shapes[var_name] = (var_type, tuple(shape))
        return sx

    def set_rhs(self, name, expr):
        self.rhs[name] = expr

    def set_expression(self, expr_name, expr):
        self.aux[expr_name] = expr
        return expr

    def setup(self):
        pass


class _Bounds(dict):
    """mpc.bounds['lower', '_u', 'u'] = array  /  mpc.bounds['lower', '_x', 'x', 3] = scalar."""


class _Template:
    """Power-index assignment tvp_template['_tvp', :, 'goal'] = value (same value at every horizon stage)."""

    def __init__(self):
        self.values = {}

    def __setitem__(self, key, value):
        assert key[0] == "_tvp" and key[1] == slice(None, None, None), key
        self.values[key[2]] = np.array(value, dtype=np.float64)

    def __getitem__(self, key):
        return self.values[key[2]]


class _Settings:
    def supress_ipopt_output(self):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None


class MPC:
    def __init__(self, model):
        self.model = model
        self.settings = _Settings()
        self.params = {}
        self.bounds = _Bounds()
        self.nl_cons = {}
        self.rterm = None
        self.objective = None
        self.tvp_fun = None
        self.x0 = None
        self.initial_guess_calls = 0

    def set_param(self, **kw):
        self.params.update(kw)

    def set_objective(self, mterm=None, lterm=None):
        self.objective = dict(mterm=mterm, lterm=lterm)

    def set_rterm(self, **kw):
        self.rterm = {k: np.array(v, dtype=np.float64) for k, v in kw.items()}

    def get_tvp_template(self):
        return _Template()

    def set_tvp_fun(self, fn):
        self.tvp_fun = fn

    def set_nl_cons(self, name, expr, ub=np.inf, **kw):
        self.nl_cons[name] = (expr, ub)
        return expr

    def setup(self):
        pass

    def set_initial_guess(self):
        self.initial_guess_calls += 1

    def make_step(self, x0):
        raise RuntimeError("no optimiser behind the recorder (IPOPT is not installed)")


class Simulator:
    def __init__(self, model):
        self.model = model

    def set_param(self, **kw):
        pass

    def get_tvp_template(self):
        return _Template()

    def set_tvp_fun(self, fn):
        pass

    def setup(self):
        pass


class StateFeedback:
    def __init__(self, model):
        self.model = model


def install_do_mpc():
    dm = types.ModuleType("do_mpc")
    dm.model = types.SimpleNamespace(Model=Model)
    dm.controller = types.SimpleNamespace(MPC=MPC)
    dm.simulator = types.SimpleNamespace(Simulator=Simulator)
    dm.estimator = types.SimpleNamespace(StateFeedback=StateFeedback)
    dm.graphics = types.SimpleNamespace(Graphics=lambda *a, **k: None)
    sys.modules["do_mpc"] = dm
    return dm
