#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<round>/ (rocprofv3 csv) into the committed summaries under profiles/.

    python tools/parse_profiles.py r01

Writes profiles/<round>_kernel_stats.csv (our kernels only), profiles/<round>_counters.json
(per-launch averages of every PMC counter per configuration) and profiles/pmc_traffic.json
(HBM bytes per launch, read by bench.py for roofline.traffic).

HBM traffic = FETCH_SIZE + WRITE_SIZE (both in KiB units; separate passes).  Per
MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced stream, i.e. half the bytes; the correction factor is CALIBRATED here on this kernel's own
access pattern (16-byte per-lane loads) from the 2^23-agent run, whose working set (2.4 GB) is far
beyond the 256 MiB Infinity Cache so it must fetch at least its algorithmic read bytes.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"profiles_{rnd}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

OURS = ("cbfqp", "mpccbf", "tracking_rollout", "tracking_coop", "tracking_select", "tracking_apply", "neighbor_kernel", "nb_bbox", "nb_count", "nb_scan", "nb_scatter",
        "nb_select", "odcbfqp", "mpclin", "mpcgn", "backupcbf", "mpcvtol", "quadtrack", "mpcdu_ms")   # manip_cbfqp matches "cbfqp"


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if any(o in k for o in OURS):
            acc[k.replace("(anonymous namespace)::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def stats(path):
    rows = []
    if not os.path.exists(path):
        return rows
    for r in csv.DictReader(open(path)):
        if any(o in r["Name"] for o in OURS):
            rows.append(r)
    return rows


# 1. kernel stats
with open(os.path.join(dst, f"{rnd}_kernel_stats.csv"), "w", newline="") as f:
    w = None
    for tag in sorted(glob.glob(os.path.join(src, "*_kernel_stats.csv"))):
        for r in stats(tag):
            r = dict(r)
            r["run"] = os.path.basename(tag).replace("_kernel_stats.csv", "")
            r["Name"] = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
            if w is None:
                w = csv.DictWriter(f, fieldnames=["run"] + [k for k in r if k != "run"])
                w.writeheader()
            w.writerow(r)

# 2. all counters
allc = {}
for path in sorted(glob.glob(os.path.join(src, "*_counter_collection.csv"))):
    allc[os.path.basename(path).replace("_counter_collection.csv", "")] = counters(path)
# what the counts belong to: a hash of the kernel sources the profiled library was built from (bench.py attaches a roofline
# only when the running tree has the same one, otherwise it says "stale")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import csrc_hash  # noqa: E402
allc["_meta"] = {"csrc_sha16": csrc_hash.csrc_sha16(ROOT), "round": rnd,
                 "note": "per-launch averages of every counter per run; bench_full_* = the default bench.py line (same seeds, same batches)"}
json.dump(allc, open(os.path.join(dst, f"{rnd}_counters.json"), "w"), indent=1, sort_keys=True)


def one(run, counter):
    d = allc.get(run, {})
    for k, c in d.items():
        if isinstance(c, dict) and counter in c:
            return c[counter]
    return None


# 3. traffic, calibrated
K, es = 8, 4
alg_read = (4 + 2 + 7 * K) * es
alg_write = (2 + K) * es + 4
cal_f = one("bigf_8388608_8_f32_f32", "FETCH_SIZE")
cal_w = one("bigw_8388608_8_f32_f32", "WRITE_SIZE")
Bcal = 8388608
fetch_factor = write_factor = None
if cal_f:
    fetch_factor = alg_read * Bcal / (cal_f * 1024.0)
if cal_w:
    write_factor = alg_write * Bcal / (cal_w * 1024.0)
# the guide's documented gfx950 correction is exactly 2.0 for wide coalesced reads; use the documented
# factor when the calibration lands near it (the calibration can only over-estimate it: extra real
# traffic lowers the apparent factor, never raises it)
ff = 2.0 if (fetch_factor and 1.6 <= fetch_factor <= 2.2) else (fetch_factor or 2.0)
wf = 1.0 if (write_factor and 0.8 <= write_factor <= 1.2) else (write_factor or 1.0)
traffic = {"_method": "bytes/launch = FETCH_SIZE*1024*fetch_factor + WRITE_SIZE*1024*write_factor (separate --pmc passes)",
           "_calibration": {"agents": Bcal, "algorithmic_read_bytes": alg_read * Bcal, "algorithmic_write_bytes": alg_write * Bcal,
                            "FETCH_SIZE_KiB": cal_f, "WRITE_SIZE_KiB": cal_w,
                            "apparent_fetch_factor": fetch_factor, "apparent_write_factor": write_factor,
                            "fetch_factor_used": ff, "write_factor_used": wf}}
for tag, B, io, comp in (("bench", 4096, "f32", "f64"), ("big_4096", 4096, "f32", "f64"), ("big_1M_f32", 1048576, "f32", "f32"),
                         ("big_1M_f64c", 1048576, "f32", "f64"), ("big_8M_f32", 8388608, "f32", "f32")):
    if tag == "bench":
        f_, w_ = one("bench_fetch", "FETCH_SIZE"), one("bench_write", "WRITE_SIZE")
    else:
        cfg = f"{B}_8_{io}_{comp}"
        f_, w_ = one("bigf_" + cfg, "FETCH_SIZE"), one("bigw_" + cfg, "WRITE_SIZE")
    if f_ is None or w_ is None:
        continue
    byt = f_ * 1024 * ff + w_ * 1024 * wf
    key = f"cbfqp_B{B}_K8_{io}" if tag == "bench" else f"cbfqp_B{B}_K8_{io}_{comp}"
    traffic[key] = byt
    traffic[key + "_detail"] = {"FETCH_SIZE_KiB": f_, "WRITE_SIZE_KiB": w_, "algorithmic_bytes": (alg_read + alg_write) * B,
                                "traffic_over_algorithmic": byt / ((alg_read + alg_write) * B)}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
vp = os.path.join(src, "valu_peak.txt")
if os.path.exists(vp):
    open(os.path.join(dst, f"{rnd}_valu_peak.txt"), "w").write(open(vp).read())
print(json.dumps(traffic, indent=1, sort_keys=True))
