"""GPU half of the code-generation guard ("The code-generation fragility: root cause").

csrc/mpc_vtol_wave.hip, csrc/mpc_vtol_ms.hip and csrc/mpc_du_ms.hip (kernel 13) are the big interior-point translation units that keep LLVM's
splitting (greedy) VGPR allocator -- the basic allocator triples their spills (mpc_vtol_ms: 45 -> 64 ms per 4096 problems; mpc_du_ms: 4.9 -> 6.8 ms).  The defect the other units are protected from by construction (copies of a live-range split placed in
front of the s_or_b64 exec of a join block) would show there as lanes losing loop-invariant values: different iterates on some
problems.  csrc/Makefile therefore builds the SAME sources a second time with the allocator that cannot split and links it into
lib/libsafe_control_hip_guard.so; this test solves the VTOL2D workload batch with both libraries (the guard one in a child process:
SAFE_CONTROL_AMD_LIB) and requires every output -- inputs, statuses, iteration counts, full plans -- to be equal BIT FOR BIT, for the
plain and the optimal-decay instantiations of the condensed kernel and of the multiple-shooting kernel (full plans x_0 .. x_N, u_0 .. u_{N-1}), and
for kernel 13 on configs[2] draws (restoration phase included) with 8 and 12 obstacle slots and on the bench draws of its four other robots, f64 and f32
storage, with the budget and its continuation launches."""
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARD = os.path.join(ROOT, "safe_control_amd", "lib", "libsafe_control_hip_guard.so")

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import safe_control_amd as sca
from safe_control_amd import workloads as W
from safe_control_amd import _lib
assert _lib.LIB_PATH == sys.argv[2], _lib.LIB_PATH
n = int(sys.argv[4])
X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("vtol", 4096, 8, seed=0))
out = {}
for io in ("f64", "f32"):
    dt = torch.float64 if io == "f64" else torch.float32
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
    for name, cls in (("plain", sca.BatchedVtolMPCCBF), ("od", sca.BatchedOptimalDecayVtolMPCCBF), ("ms", lambda io_dtype: sca.BatchedVtolMSMPCCBF(io_dtype=io_dtype, fallback=False)),
                      ("odms", lambda io_dtype: sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype=io_dtype, fallback=False))):
        ctl = cls(io_dtype=io)
        m = n if name != "odms" else n // 4                     # (both instantiations of the multiple-shooting kernel: the optimal-decay one on a quarter of the batch)
        r = ctl.solve(t(X[:m]), t(up[:m]), t(goal[:m]), t(obs[:m]), want_z=True)
        torch.cuda.synchronize()
        for k, a in enumerate(r):
            if a is not None:
                out[f"{io}/{name}/{k}"] = a.cpu().numpy()
    # kernel 13 (csrc/mpc_du_ms.hip: DynamicUnicycle2D, multiple shooting): configs[2] draws, one in ten without a feasible point -- those run
    # through the restoration phase --, and a batch with 12 obstacle slots
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25}
    for name, K in (("dums", 8), ("dums12", 12)):
        Xd, upd, gd, od = (a[: 2 * n] for a in W.mpc_family_batch("du", 2 * n, K, seed=0))
        r = sca.BatchedMSMPCCBF(spec, io_dtype=io).solve(t(Xd), t(upd), t(gd), t(od), want_plan=True)
        torch.cuda.synchronize()
        for k, a in enumerate(r):
            out[f"{io}/{name}/{k}"] = a.cpu().numpy()
    # ... and its other robots (the same translation unit, the same allocator): Unicycle2D, SingleIntegrator2D, DoubleIntegrator2D, KinematicBicycle2D
    for fam in ("uni", "si", "di", "kb"):
        Xf, upf, gf, of = (a[:n] for a in W.mpc_family_batch(fam, n, 8, seed=0))
        r = sca.BatchedMSMPCCBF({"model": W.MPC_FAMILIES[fam]}, io_dtype=io, max_iter=150).solve(t(Xf), t(upf), t(gf), t(of), want_plan=True)
        torch.cuda.synchronize()
        for k, a in enumerate(r):
            out[f"{io}/dums_{fam}/{k}"] = a.cpu().numpy()
np.savez(sys.argv[3], **out)
"""


def run(lib, out, n):
    env = dict(os.environ, SAFE_CONTROL_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, lib, out, str(n)], env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def test_wave_kernel_equals_its_build_with_the_allocator_that_cannot_split(tmp_path):
    if not os.path.exists(GUARD):
        pytest.fail(f"{GUARD} is missing: `make -C safe_control_amd/csrc` builds it next to the library")
    shipped = os.path.join(ROOT, "safe_control_amd", "lib", "libsafe_control_hip.so")
    n = 512
    a = run(shipped, str(tmp_path / "a.npz"), n)
    b = run(GUARD, str(tmp_path / "b.npz"), n)
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 40
    assert (a["f64/dums/1"] == 1).mean() > 0.05                          # (kernel 13's restoration phase is in the comparison)
    for k in a.files:
        x, y = a[k], b[k]
        assert x.dtype == y.dtype and x.shape == y.shape
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (k, int((x != y).sum()))
    st = a["f64/plain/1"]
    assert (st == 0).mean() > 0.9 and a["f64/plain/2"].max() > 100        # the batch does reach the continuation launches
