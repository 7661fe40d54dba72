#!/usr/bin/env python3
"""Print the markdown table of profiles/<round>_kernel_stats.csv (run, kernel, calls, average / min per launch) so that
profiles/README.md quotes the committed CSV and nothing else:   python tools/profiles_table.py r02"""
import csv, os, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{rnd}_kernel_stats.csv")
print("| run | kernel | calls | avg per launch | min |")
print("|---|---|---|---|---|")
for r in csv.DictReader(open(path)):
    a, m = float(r["AverageNs"]), float(r["MinNs"])
    f = (lambda v: f"{v / 1e6:.2f} ms") if a >= 1e6 else (lambda v: f"{v / 1e3:.2f} µs")
    print(f"| `{r['run']}` | `{r['Name'].replace('void sc::', '')}` | {r['Calls']} | {f(a)} | {f(m)} |")
