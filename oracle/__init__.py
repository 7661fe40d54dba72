"""CPU oracle for the batched CBF-QP / MPC-CBF hot path.

TEST INFRASTRUCTURE ONLY.  This package is a float64 CPU restatement of the
reference algorithm (tkkim-robot/safe_control: position_control/cbf_qp.py,
position_control/mpc_cbf.py, robots/*.py, tracking.py).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, call, link or execute anything under ``oracle/`` -- and there only as
the checker, never as the thing measured as the product or shipped.  The
product path (``safe_control_amd``) never imports this package and has no CPU
fallback: it fails loudly if the HIP library is missing.

Parity pinning (see DESIGN.md "Oracle"):

* QP *data* (f, g, step, nominal_input, agent_barrier, assembled A/b rows,
  obstacle selection) is pinned against the reference's own numpy functions,
  imported in the build container with a ``casadi``/``shapely``/``cvxpy`` type
  stub (tests/golden/make_golden.py); the resulting vectors are committed
  under tests/golden/.
* The CBF-QP *solution* u* comes from GUROBI through cvxpy in the reference
  (position_control/cbf_qp.py:190); neither is installable here.  The QP is
  strictly convex (identity Hessian) so u* is unique; the oracle finds it by
  exact active-set enumeration in float64 and is cross-checked with
  scipy SLSQP.  At the GUROBI boundary: parity unpinned by the reference,
  pinned by uniqueness of the minimiser.
* The MPC-CBF NLP is solved by do-mpc -> casadi -> IPOPT in the reference
  (position_control/mpc_cbf.py:384); none are available and the NLP is
  non-convex: **parity unpinned**.  The oracle restates the NLP (cost,
  dynamics, DT-CBF constraints, bounds) and solves it with its own float64
  SQP from the same constant initial guess; tests check feasibility, KKT
  residual and cost against it and against scipy.
"""
