"""GPU: launch time of one family's 4096-problem bench batch through variant libraries (tools/build_variants.sh), each in a child process.
    python3 tools/time_variants.py FAMILY exp_libs/lib_A.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--child":
    fam, lib = sys.argv[2], sys.argv[3]
    from safe_control_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(lib)
    import numpy as np, torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exp_tail as E
    ctl, arrs = E.make(fam, 100)
    ms, st, it = E.run(ctl, arrs, reps=3)
    print(f"{fam:7s} {os.path.basename(lib):22s} {ms:9.3f} ms   status 0/1/2 = {[int((st == s).sum()) for s in (0, 1, 2)]}  it mean {it.mean():.2f} max {it.max()}", flush=True)
    sys.exit(0)
fam = sys.argv[1]
for lib in sys.argv[2:]:
    subprocess.call([sys.executable, os.path.abspath(__file__), "--child", fam, lib])
