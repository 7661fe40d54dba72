"""ONE kind of dispatch for a counter pass: a family's 4096-problem batch filled with copies of its median-iteration optimal problem
(bench.py: budget_note "uniform_batch"), one launch stopped at 100 iterations.  The draw is found by a first solve of the real batch with a
controller whose kernel name differs in nothing, so the pass sees two dispatches of the same kernel: the tool prints which is which
(the uniform one is the LAST dispatch).   python3 tools/prof_uniform.py FAMILY [reps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exp_tail as E

fam = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctl, arrs = E.make(fam, 100)
ctl.iter_slices, ctl.classify_first = (), False
ms, st, it = E.run(ctl, arrs, reps=1)                       # dispatches 1 - 2: the real batch
opt = np.flatnonzero(st == 0)
j = int(opt[np.argsort(it[opt])[len(opt) // 2]])
rep = [np.repeat(a[j:j + 1], a.shape[0], axis=0) for a in arrs]
msu, stu, itu = E.run(ctl, rep, reps=reps)                  # the uniform batch: reps + 1 dispatches
print(f"{fam}: real batch {ms:.3f} ms (mean {it.mean():.2f} iterations, max {it.max()}); uniform batch of draw {j}: {msu:.3f} ms, {int(itu[0])} iterations each, "
      f"all equal {bool((itu == itu[0]).all())}")
