"""GPU: random continuation schedules (caps, order, classify_first) against the uninterrupted solve, bit for bit, for every family.
    python3 tools/exp_fuzz_slices.py [schedules per family] [problems]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mpc_slices_gpu as T

n_sched = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 192
rng = np.random.default_rng(2026)
bad = 0
for fam in T.FAMILIES:
    arrs = T.batch(fam, B, seed=7)
    ref = T.solve(T.make(fam, max_iter=300), arrs)
    for s in range(n_sched):
        k = int(rng.integers(1, 9))
        caps = tuple(sorted(set(int(c) for c in rng.integers(1, max(4, int(ref[2].max())), size=k))))
        kw = dict(iter_slices=caps, order=bool(rng.integers(0, 2)), classify_first=bool(rng.integers(0, 2)))
        got = T.solve(T.make(fam, max_iter=300, **kw), arrs)
        try:
            T.same(ref, got, f"{fam} {kw}")
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e)
    print(f"{fam}: {n_sched} random schedules on {B} problems (iterations up to {int(ref[2].max())}, status 0/1/2 = "
          f"{[int((ref[1] == q).sum()) for q in (0, 1, 2)]}): bitwise equal" if bad == 0 else f"{fam}: mismatches so far {bad}", flush=True)
sys.exit(1 if bad else 0)
