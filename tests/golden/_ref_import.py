"""Import shim used ONLY by tests/golden/make_golden.py in the build container.

The reference (/root/reference) top-level-imports casadi, shapely, cvxpy and
do_mpc, none of which are installed here.  Its *numpy* halves (dynamics,
barrier callbacks, CBF-QP row assembly, obstacle selection, the control_step
state machine) run fine once those names resolve, so this module registers
inert stand-in modules in ``sys.modules`` and maps the package name
``safe_control`` onto the reference checkout the way its pyproject.toml does
(pyproject.toml:22-24).  Nothing here is a re-implementation of those
libraries and nothing from the reference is copied: the stubs only let the
reference's own source files import.  The GPU box has no /root/reference;
nothing under tests/ imports this file at test time.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Anything:
    """Absorbs any construction / attribute / call / operator and returns itself."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return self

    def _binop(self, *a, **k):
        return self

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = _binop
    __matmul__ = __rmatmul__ = __truediv__ = __rtruediv__ = _binop
    __le__ = __ge__ = __lt__ = __gt__ = __neg__ = __pow__ = _binop
    __getitem__ = _binop

    def __iter__(self):
        return iter(())


class _Param(_Anything):
    """cvxpy.Parameter / Variable stand-in: just holds ``.value``."""

    def __init__(self, shape=None, value=None, **k):
        object.__setattr__(self, "value", value)
        object.__setattr__(self, "shape", shape)

    def __getattr__(self, name):
        raise AttributeError(name)


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True   # /root/reference must stay untouched

    class SX(_Anything):
        pass

    class MX(_Anything):
        pass

    class DM(_Anything):
        pass

    _module("casadi", SX=SX, MX=MX, DM=DM, pi=3.141592653589793)

    geom = _module("shapely.geometry", Polygon=_Anything, Point=_Anything,
                   LineString=_Anything, MultiPolygon=_Anything)
    ops = _module("shapely.ops", unary_union=_Anything())
    val = _module("shapely.validation", explain_validity=_Anything())
    _module("shapely", geometry=geom, ops=ops, validation=val, is_valid_reason=_Anything())

    _module("cvxpy", Variable=_Param, Parameter=_Param, Minimize=_Anything,
            Problem=_Anything, sum_squares=_Anything(), abs=_Anything(), square=_Anything(),
            GUROBI="GUROBI", OSQP="OSQP", SCS="SCS")
    dm = _module("do_mpc")
    for sub in ("model", "controller", "simulator", "estimator", "graphics"):
        setattr(dm, sub, _Anything())

    import matplotlib
    matplotlib.use("Agg")

    pkg = types.ModuleType("safe_control")
    pkg.__path__ = [REFERENCE_ROOT]
    sys.modules["safe_control"] = pkg
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
