#!/usr/bin/env python3
"""Benchmark of the batched CBF-QP hot path on MI355X (BASELINE.json metric:
"QP solves/sec (batched agents)").

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: one fused kernel launch
that assembles the CBF rows of every agent and solves every QP.  The default
workload is BASELINE.json configs[1]: 4096 DynamicUnicycle2D agents x 8
circular obstacles each (seeded synthetic inputs, resident in HBM before the
timed region).  With N > 1 (launched by torch.distributed.run, one rank per
GPU) every rank owns its own 4096-agent shard -- weak scaling, no data-path
collective -- and the job time is the MAX over ranks.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      HBM roofline of the fused kernel on this workload (algorithmic bytes / launch time)
  cpu_baseline  the oracle's C restatement timed on this box's host cores (N = 1 only)
  sweep         the same kernel at larger batches, where the HBM roofline is the binding limit
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6300.0   # what a streaming kernel reaches on this part (MI355X_MICROARCH.md: "8 TB/s peak (spec); ~6.3 TB/s achievable")
BYTES_IN_CBFQP = lambda K, es: (4 + 2 + 7 * K) * es            # X + u_ref + obs rows   # noqa: E731
BYTES_OUT_CBFQP = lambda K, es: (2 + K) * es + 4               # u + h + status(int32)   # noqa: E731


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--agents", type=int, default=4096, help="agents per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--obstacles", type=int, default=8)
    ap.add_argument("--compute", choices=["f32", "f64"], default="f64", help="arithmetic type inside the kernel")
    ap.add_argument("--io", choices=["f32", "f64"], default="f32", help="storage type of states/obstacles/outputs")
    ap.add_argument("--eager", action="store_true", help="launch each step from Python instead of one hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--workload", choices=["cbf_qp", "mpc_cbf", "kb_c3bf", "hetero_fleet"], default="cbf_qp",
                    help="cbf_qp = BASELINE configs[1] (default, the headline metric); mpc_cbf = configs[2]; "
                         "kb_c3bf = configs[3]: 16384 KinematicBicycle2D C3BF agents in total (strong scaling), each "
                         "taking its 16 nearest other agents as moving obstacles after an all-gather of the states; "
                         "hetero_fleet = configs[4] (extension, SURVEY 8d): a 65536-agent fleet, half Unicycle2D and half "
                         "Quad3D, optimal-decay MPC-CBF with horizon 20 and 8 superellipsoid obstacles each, sharded over "
                         "the ranks (strong scaling), the two model kernels on two HIP streams")
    ap.add_argument("--plain-fleet", action="store_true",
                    help="hetero_fleet: plain MPCCBF of both models on circular obstacles (round 1's variant)")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--max-iter", type=int, default=0,
                    help="hetero_fleet: iteration limit of the interior point (0: the classes' default, IPOPT's 3000; 100: rounds 1 - 3)")
    ap.add_argument("--no-mpc", action="store_true", help="skip the short MPC-CBF leg of the default run")
    ap.add_argument("--no-limit100", action="store_true",
                    help="interior-point legs: skip the extra timing of the round-3 configuration (one launch, 100 iterations); the counter "
                         "passes of tools/collect_profiles.sh use it so that every dispatch of a leg belongs to a budget solve")
    return ap.parse_args()


def cpu_baseline(X, u_ref, obs, seconds):
    """Oracle (C restatement, kind "port") on the host cores: bounded sample of the same workload."""
    import numpy as np
    from oracle import c_oracle, cbf_qp as ocbf, robots as R
    spec = R.default_spec(R.MODEL_DU)
    spec.update(a_max=1.0, w_max=0.5, radius=0.25)
    cp = ocbf.default_cbf_param(R.MODEL_DU)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 64, X.shape[0] // 64))           # >= 64 agents per thread
    out = {}
    for label, nt, budget in (("all_cores", threads, seconds * 0.6), ("one_core", 1, seconds * 0.4)):
        c_oracle.cbfqp_batch(R.MODEL_DU, X, u_ref, obs, spec, cp, n_threads=nt)      # warm
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            c_oracle.cbfqp_batch(R.MODEL_DU, X, u_ref, obs, spec, cp, n_threads=nt)
            n += 1
        dt = time.perf_counter() - t0
        out[label] = (n * X.shape[0] / dt, nt, n)
    # B0 of BASELINE.md section 4: the numpy oracle called agent by agent from Python -- the closest stand-in for the
    # reference's per-call cost (its own stack, cvxpy + GUROBI per control step, is not installable here)
    n_py, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min(3.0, 0.25 * seconds) and n_py < X.shape[0]:
        ocbf.solve(R.MODEL_DU, X[n_py], u_ref[n_py], obs[n_py], spec, cp, num_obs=obs.shape[1])
        n_py += 1
    py_rate = n_py / (time.perf_counter() - t0)
    best = "all_cores" if out["all_cores"][0] >= out["one_core"][0] else "one_core"
    v, nt, n = out[best]
    return {"value": v, "unit": "solves/s", "cores": nt, "kind": "port",
            "sample": f"{n} passes over the same {X.shape[0]} x {obs.shape[1]} batch, oracle/c/cbfqp_oracle.c (f64, exact active-set enumeration), OpenMP {nt} threads",
            "one_core_value": out["one_core"][0], "all_cores_value": out["all_cores"][0],
            "python_per_agent_loop_value": py_rate, "python_per_agent_loop_sample": f"first {n_py} agents, oracle/cbf_qp.py (numpy float64), 1 thread",
            "all_cores_threads": out["all_cores"][1]}


NO_LIMIT100 = False            # --no-limit100
LAUNCHES_PER_BUDGET_SOLVE = 3  # classify + cap 100 + rest: the counter profile holds per-dispatch averages of a kernel
VALU_PEAK_GIPS = 519.0   # measured sustained f64 VALU issue rate of the chip, G wave-instructions/s (tools/micro/valu_peak.hip,
                         # profiles/r02_valu_peak.txt: 0.507 G/s per SIMD at four waves per SIMD; 1024 SIMDs x 2.4 GHz / 4 = 614 on paper)


def csrc_sha16():
    """Hash of the kernel sources of this tree as code (comments and blank lines stripped: tools/csrc_hash.py;
    tools/parse_profiles.py stores the same one with the counter profile)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import csrc_hash
    finally:
        sys.path.pop(0)
    return csrc_hash.csrc_sha16(ROOT)


F64_VECTOR_PEAK_TFLOPS = 78.6                                   # MI355X data sheet, vector f64 (MI355X_MICROARCH.md)


def valu_roofline(run, kernel_substr, kernel_ms, note=None, launches=1):
    """VALU-issue roofline of an interior-point kernel: VALU wave-instructions per launch -- the SQ_INSTS_VALU counter of
    the same launch configuration, collected with rocprofv3 --pmc and committed under profiles/ (the count is a property
    of the batch: same seed, same iterates) -- over this run's measured launch time, against the measured issue peak.
    Issue slots are not useful work (they count spill moves, DPP moves and idle lanes), so two more figures ride along when the
    profile has them: lane utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), and the f64 flops of the launch
    (lanes x (2 FMA + ADD + MUL) + 512 per MFMA op) against the 78.6 TFLOP/s vector peak.  The profile is used only when it
    was taken from the kernel sources of this tree (csrc hash); otherwise the entry says "stale"."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_counters.json")
        if not os.path.exists(path):
            continue
        try:
            allc = json.load(open(path))
        except Exception:
            continue
        d = allc.get(run, {})
        meta = allc.get("_meta", {})
        for k, c in d.items():
            name = k.replace("void sc::", "")
            if (name == kernel_substr or kernel_substr in k) and "SQ_INSTS_VALU" in c:
                insts = c["SQ_INSTS_VALU"] * launches                   # per-dispatch average x dispatches of one solve
                ach = insts / (kernel_ms * 1e-3) / 1e9
                out = {"bound": "valu_issue", "achieved": ach, "peak": VALU_PEAK_GIPS, "unit": "G wave-instr/s", "frac": ach / VALU_PEAK_GIPS,
                       "traffic": None, "kernel": name, "kernel_us": 1e3 * kernel_ms,
                       "valu_instructions_per_launch": insts, "source": f"profiles/{rnd}_counters.json:{run}",
                       "stale": meta.get("csrc_sha16") != csrc_sha16()}
                if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_ACTIVE_INST_LDS"):
                    r_ = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"]
                    if r_ <= 1.0:                                           # (a kernel with a handful of LDS instructions gives a meaningless ratio)
                        out["lds_bank_conflict_fraction"] = r_
                lane = allc.get(run.replace("_sq", "_lane"), {}).get(k)
                flop = allc.get(run.replace("_sq", "_flop"), {}).get(k)
                if lane and lane.get("SQ_ACTIVE_INST_VALU"):
                    out["lane_utilisation"] = lane["SQ_THREAD_CYCLES_VALU"] / (64.0 * lane["SQ_ACTIVE_INST_VALU"])
                if flop and "lane_utilisation" in out:
                    vec = 2.0 * flop.get("SQ_INSTS_VALU_FMA_F64", 0.0) + flop.get("SQ_INSTS_VALU_ADD_F64", 0.0) + flop.get("SQ_INSTS_VALU_MUL_F64", 0.0)
                    fl = launches * (64.0 * out["lane_utilisation"] * vec + 512.0 * flop.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0))
                    out["f64_flops_per_launch"] = fl
                    out["f64_tflops"] = fl / (kernel_ms * 1e-3) / 1e12
                    out["f64_vector_peak_frac"] = out["f64_tflops"] / F64_VECTOR_PEAK_TFLOPS
                if note:
                    out["note"] = note
                return out
    return None


def work_level(r, ub, mean_it, B):
    """Issue rate at the WORK level (no tail): VALU instructions per iteration and problem from the profiled batch (instructions /
    sum of iterations), times the iterations of the uniform batch (budget_note: copies of the median-iteration problem, one launch),
    over that batch's launch time."""
    if not (isinstance(r, dict) and isinstance(ub, dict) and ub.get("iterations") and mean_it and B):
        return
    per_it = r["valu_instructions_per_launch"] / (mean_it * B)
    ach = per_it * ub["iterations"] * B / (ub["kernel_ms"] * 1e-3) / 1e9
    r["work_level"] = {"achieved": ach, "frac": ach / VALU_PEAK_GIPS, "us_per_iteration_of_the_batch": 1e3 * ub["kernel_ms"] / ub["iterations"],
                       "note": "uniform batch: copies of the median-iteration problem, one launch"}


def with_roofline(res, kernel_substr, launches=LAUNCHES_PER_BUDGET_SOLVE):
    """Attach the VALU-issue roofline of an interior-point leg when the committed counter profile holds the kernel
    (profiles/r04_counters.json: run bench_full_sq = the default bench command with --no-limit100 under rocprofv3 --pmc, same
    batches; a budget solve is `launches` dispatches of the kernel)."""
    r = valu_roofline("bench_full_sq", kernel_substr, res["kernel_ms"], launches=launches)
    if r is not None:
        res["roofline"] = r
        try:
            B = int(res["workload"].split("-")[0].split()[0]) if res.get("workload", " ")[0].isdigit() else None
        except Exception:
            B = None
        work_level(r, res.get("uniform_batch"), res.get("mean_ipm_iterations"), B or res.get("agents"))
    return res


def budget_note(make_ctl, args, steps, ms_full, st_full, it_full):
    """The interior-point legs run the reference solver's budget (IPOPT's max_iter = 3000, mpc_cbf.py:163-173) as continuation launches
    (first cap 100, classify-only pre-pass).  Beside that figure: the same batch with the round-3 limit of 100 iterations in one launch,
    and how many problems needed more."""
    import torch
    base = {"budget": 3000, "launches": "classify + cap 100 + rest", "max_ipm_iterations": int(it_full.max().item()),
            "inaccurate_fraction": float((st_full == 2).double().mean().item()), "beyond_100_iterations": int((it_full > 100).sum().item())}
    if NO_LIMIT100:
        return base
    ctl = make_ctl(max_iter=100, iter_slices=(), classify_first=False)
    out = ctl.solve(*args)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = ctl.solve(*args)
    e1.record()
    torch.cuda.synchronize()
    st = out[1]
    res = {**base, "one_launch_limit_100": {"kernel_ms": e0.elapsed_time(e1) / steps, "optimal_fraction": float((st == 0).double().mean().item()),
                                            "inaccurate_fraction": float((st == 2).double().mean().item())}}
    # The work without the tail: the batch filled with copies of ONE problem -- the optimal solve with the median iteration count -- so
    # that every wave runs the same number of iterations and the launch time is work / machine, not longest solve x lone-wave latency.
    try:
        opt = torch.nonzero(st_full == 0).flatten()
        if opt.numel() >= 8:
            its = it_full[opt]
            j = int(opt[torch.argsort(its)[its.numel() // 2]].item())
            B = int(st_full.shape[0])
            rep = tuple((a[j:j + 1].expand(B, *a.shape[1:]).contiguous() if a.shape[0] == B else a) for a in args)
            o2 = ctl.solve(*rep)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                o2 = ctl.solve(*rep)
            e1.record()
            torch.cuda.synchronize()
            res["uniform_batch"] = {"copies_of_draw": j, "iterations": int(o2[2][0].item()), "kernel_ms": e0.elapsed_time(e1) / steps,
                                    "all_equal": bool((o2[2] == o2[2][0]).all().item())}
    except Exception as e:                                           # an extra figure never takes the leg down
        res["uniform_batch"] = {"error": repr(e)[:120]}
    return res


def mpc_cpu_baseline(Xn, goal, on, N, seconds):
    """BASELINE configs[2] on the host cores: the multiple-shooting solve compiled for the CPU (oracle/c/mpc_du_ms_cpu.cpp: the same
    algorithm as kernel 13 -- oracle/ms_ipopt.py's, which tests/test_oracle_c.py holds it to --, the wavefront's lanes as fibers of one
    thread, OpenMP over problems) on a bounded sample of the same batch, one core and all cores; beside it the numpy oracle on one core."""
    import numpy as np
    from oracle import c_oracle, ms_ipopt as MS
    up = np.zeros((Xn.shape[0], 2))
    nt_all = c_oracle.load_ms().du_ms_cpu_num_threads()
    out = {}
    for label, nt, share in (("one_core", 1, 0.3), ("all_cores", 0, 0.45)):
        n, t0, chunk = 0, time.perf_counter(), (32 if nt == 1 else 64 * max(1, nt_all))
        c_oracle.du_ms_cpu_batch(Xn[:8], up[:8], goal[:8], on[:8], horizon=N, n_threads=nt)      # (thread pool and page faults out of the way)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < share * seconds and n < Xn.shape[0]:
            m = min(chunk, Xn.shape[0] - n)
            c_oracle.du_ms_cpu_batch(Xn[n:n + m], up[n:n + m], goal[n:n + m], on[n:n + m], horizon=N, n_threads=nt)
            n += m
        out[label] = (n / (time.perf_counter() - t0), nt_all if nt == 0 else 1, n)
    n_py, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 0.2 * seconds and n_py < Xn.shape[0]:
        MS.solve(MS.du_model(), Xn[n_py], up[n_py], goal[n_py], on[n_py], N=N, opts=MS.KERNEL_PROFILE)
        n_py += 1
    py_rate = n_py / (time.perf_counter() - t0)
    best = "all_cores" if out["all_cores"][0] >= out["one_core"][0] else "one_core"
    v, nt, n = out[best]
    return {"value": v, "unit": "solves/s", "cores": nt, "kind": "port",
            "sample": f"first {n} problems, oracle/c/mpc_du_ms_cpu.cpp (oracle/ms_ipopt.py compiled), OpenMP {nt} threads",
            "one_core_value": out["one_core"][0], "all_cores_value": out["all_cores"][0],
            "python_oracle_value": py_rate, "python_oracle_sample": f"first {n_py} problems, oracle/ms_ipopt.py (numpy float64), 1 thread"}


def mpc_leg(dev, B, K, N, steps, warmup, seed=0, cpu_seconds=0.0):
    """BASELINE configs[2]: B DynamicUnicycle2D agents, MPC-CBF horizon N, K obstacles, u_prev = 0 -- in the reference's own formulation
    (do-mpc's multiple shooting under IPOPT's algorithm, restoration phase included: kernel 13, csrc/mpc_du_ms.hip; the default since
    round 6).  `condensed`: the same batch on kernel 3 (single shooting, l1-merit interior point), the formulation of rounds 1 - 5,
    selectable with robot_spec['mpc_formulation'] = 'condensed'."""
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    mk = lambda **kw: sca.BatchedMPCCBF(dict(spec), io_dtype="f32", horizon=N, **kw)   # noqa: E731
    ms = sca.BatchedMSMPCCBF(dict(spec), io_dtype="f32", horizon=N, check_circles=False)
    cond = mk()
    Xn, goal, un, on = W.du_cbfqp_batch(B, K, seed=seed)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob = t(Xn), t(goal), t(on)
    up = torch.zeros((B, 2), dtype=torch.float32, device=dev)

    def timed(ctl, n_out):
        out = (torch.empty((B, 2), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
               torch.empty((B,), dtype=torch.int32, device=dev)) + ((None,) if n_out == 4 else ())
        for _ in range(max(1, warmup)):
            ctl.solve(X, up, g, ob, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            ctl.solve(X, up, g, ob, out=out)
        e1.record()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0, e0.elapsed_time(e1) / steps

    out, wall, ms_ms = timed(ms, 3)
    st, it = out[1], out[2]
    outc, wallc, ms_c = timed(cond, 4)
    stc, itc = outc[1], outc[2]
    nbytes = ((4 + 2 + 2 + 7 * K) * 4 + 2 * 4 + 4 + 4) * B
    extra = {"cpu_baseline": mpc_cpu_baseline(Xn, goal, on, N, cpu_seconds)} if cpu_seconds > 0 else {}
    both = (st == 0) & (stc == 0)
    condensed = {"kernel": "mpccbf_kernel<10, 8> (kernel 3: single shooting, l1-merit interior point)", "value": B * steps / wallc, "kernel_ms": ms_c,
                 "optimal_fraction": float((stc == 0).double().mean().item()), "infeasible_fraction": float((stc == 1).double().mean().item()),
                 "mean_ipm_iterations": float(itc.double().mean().item()),
                 "same_status_fraction": float((st == stc).double().mean().item()),
                 "same_u0_where_both_optimal_fraction": float(((out[0] - outc[0]).abs().amax(dim=1) <= 1e-4)[both].double().mean().item()) if bool(both.any()) else None,
                 "u0_differs_by_more_than_1e-3_where_neither_is_optimal_fraction":
                     float(((out[0] - outc[0]).abs().amax(dim=1) > 1e-3)[(st != 0) & (stc != 0)].double().mean().item()) if bool(((st != 0) & (stc != 0)).any()) else None}
    condensed.update(budget_note(mk, (X, up, g, ob), steps, ms_c, stc, itc))
    if (B, K, N, seed) == (4096, 8, 10, 0):
        rlc = valu_roofline("mpc_sq", "mpccbf_kernel<10, 8>", ms_c, launches=LAUNCHES_PER_BUDGET_SOLVE)
        if rlc:
            work_level(rlc, condensed.get("uniform_batch"), float(itc.double().mean().item()), B)
            condensed["roofline"] = rlc
        rl = valu_roofline("mpc_sq", "mpcdu_ms_kernel<float, 0,", ms_ms, launches=1, note="one wave per problem, four lanes per stage; 512 registers: one wave "
                           "per SIMD = 1024 resident problems; the launch ends with its slowest problem (94 iterations, most of them inside the restoration phase, "
                           "against a mean of 17.6 at ~40 us per iteration)")
        if rl:
            extra["roofline"] = rl
    return {**extra, "workload": f"{B}-agent batch DynamicUnicycle2D MPC-CBF, horizon N={N}, {K} obstacles (BASELINE configs[2])",
            "formulation": "multiple shooting under IPOPT's filter interior point, restoration phase in the kernel (do-mpc's NLP: mpc_cbf.py:162-174,366-402)",
            "kernel": "mpcdu_ms_kernel<float, 0> (kernel 13)",
            "value": B * steps / wall, "unit": "solves/s", "steps": steps, "kernel_ms": ms_ms,
            "dtype": "f64", "storage": "f32", "budget": 3000, "launches": "one",
            "optimal_fraction": float((st == 0).double().mean().item()),
            "infeasible_fraction": float((st == 1).double().mean().item()),
            "inaccurate_fraction": float((st == 2).double().mean().item()),
            "mean_ipm_iterations": float(it.double().mean().item()), "max_ipm_iterations": int(it.max().item()),
            "achieved_GBs": nbytes / (ms_ms * 1e-3) / 1e9, "algorithmic_bytes_per_solve": nbytes // B,
            "condensed": condensed}


def pipelined_leg(ctl, dev, X, ur, ob, K, steps):
    """The headline workload with TWO independent batches in flight: even steps on one HIP stream, odd steps on another,
    each with its own input / output buffers, both captured in one hipGraph (fork / join).  One 4096-agent launch is 512
    waves on 1024 SIMDs and a launch-bound 4 us; a second, independent batch fills the other half of the chip.  Reported
    beside the headline, never as it: the headline keeps one batch per step, strictly in sequence."""
    import torch
    B = X.shape[0]
    td = X.dtype
    bufs = []
    for _ in range(2):
        bufs.append((X.clone(), ur.clone(), ob.clone(),
                     (torch.empty((B, 2), dtype=td, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
                      torch.empty((B, K), dtype=td, device=dev))))
    s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    s0.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s0):
        for b in bufs:
            ctl.solve(b[0], b[1], b[2], out=b[3])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s0):
            s1.wait_stream(s0)                                   # fork
            for k in range(steps):
                b = bufs[k & 1]
                with torch.cuda.stream(s0 if (k & 1) == 0 else s1):
                    ctl.solve(b[0], b[1], b[2], out=b[3])
            s0.wait_stream(s1)                                   # join
    torch.cuda.current_stream(dev).wait_stream(s0)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    dt_ = time.perf_counter() - t0
    same = bool(torch.equal(bufs[0][3][0].nan_to_num(), bufs[1][3][0].nan_to_num()))
    return {"workload": f"{B}-agent CBF-QP batches, two independent batches in flight (two HIP streams inside one hipGraph)",
            "steps": steps, "us_per_step": 1e6 * dt_ / steps, "solves_per_s": B * steps / dt_, "outputs_identical": same}


def closed_loop_mpc_leg(dev, B=4096, T=20):
    """Closed loop with the reference's default position controller (examples/test_tracking.py:15, --algo mpc_cbf):
    per step select -> one MPC-CBF launch for the batch -> apply, on the 14-circle scene."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    obs = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0], [7.0, 7.0, 3.0],
                    [4.0, 3.5, 1.5], [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6], [11.0, 5.0, 0.8],
                    [13.5, 11.0, 0.6], [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
    rng = np.random.default_rng(0)
    P = rng.uniform(0.5, 13.5, (4 * B, 2))                                       # starts at least 0.6 m clear of every circle
    clear = (np.hypot(P[:, None, 0] - obs[None, :, 0], P[:, None, 1] - obs[None, :, 1]) - obs[None, :, 2]).min(axis=1) > 0.85
    P = P[clear][:B]
    X0 = np.column_stack([P, rng.uniform(-np.pi, np.pi, B), rng.uniform(0, 1, B)])
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25, "num_constraints": 8}
    ctl = sca.BatchedTrackingController(X0, spec, controller_type={"pos": "mpc_cbf"}, obs=obs, io_dtype="f32", device=str(dev))
    ctl.set_waypoints(np.array([[2.0, 2.0], [2.0, 12.0], [12.0, 12.0], [12.0, 2.0]]))
    ctl.control_step(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctl.control_step(T)
    torch.cuda.synchronize()
    dt_ = time.perf_counter() - t0
    return {"workload": f"{B} DynamicUnicycle2D agents x {T} closed-loop control steps, MPC-CBF (N=10, 8 nearest obstacles) "
                        "as position controller, 14 shared obstacles",
            "ms_per_control_step": 1e3 * dt_ / T, "agent_steps_per_s": B * T / dt_,
            "running": int((ctl.ret == 0).sum().item()), "dtype": "f64", "storage": "f32"}


def od_mpc_leg(dev, B=4096, K=8, N=10, steps=2, seed=0):
    """Optimal-decay MPC-CBF on the config-3 batch: two decay variables per stage."""
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    ctl = sca.BatchedOptimalDecayMPCCBF(dict(spec), io_dtype="f32", horizon=N)
    Xn, goal, un, on = W.du_cbfqp_batch(B, K, seed=seed)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob = t(Xn), t(goal), t(on)
    up = torch.zeros((B, 2), dtype=torch.float32, device=dev)
    u, rho, st, it = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        u, rho, st, it = ctl.solve(X, up, g, ob)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return with_roofline({"workload": f"{B}-agent batch DynamicUnicycle2D optimal-decay MPC-CBF, horizon N={N}, {K} obstacles",
            "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32",
            "optimal_fraction": float((st == 0).double().mean().item()),
            "mean_ipm_iterations": float(it.double().mean().item()),
            "max_decay_deviation": float((rho - 1.0).abs().max().item())}, f"odmpccbf_kernel<{N}>", launches=1)


def manip_leg(dev, B=4096, K=3, steps=20, seed=0):
    """Manipulator2D CBF-QP: B three-joint arms, K circular obstacles each in the arm's workspace
    (25 link circles per obstacle -> 25 K rows, row cap 150 as in tracking.py:134-138), one QP per wavefront."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    rng = np.random.default_rng(seed)
    X = rng.uniform(-np.pi, np.pi, (B, 3))
    ur = rng.uniform(-2.5, 2.5, (B, 3))
    rho, phi = rng.uniform(0.8, 3.8, (B, K)), rng.uniform(-np.pi, np.pi, (B, K))
    obs = np.zeros((B, K, 7))
    obs[..., 0], obs[..., 1], obs[..., 2] = rho * np.cos(phi), rho * np.sin(phi), rng.uniform(0.15, 0.5, (B, K))
    ctl = sca.BatchedManipulatorCBFQP({"model": "Manipulator2D", "w_max": 2.0, "radius": 0.25}, io_dtype="f32", num_rows=150)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    tX, tu, to = t(X), t(ur), t(obs)
    u, st, h = ctl.solve(tX, tu, to)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        u, st, h = ctl.solve(tX, tu, to)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    rows = min(150, 25 * K)
    return {"workload": f"{B} Manipulator2D arms, {K} obstacles each ({rows} CBF rows + 6 box rows, 3 inputs)",
            "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32",
            "optimal_fraction": float((st == 0).double().mean().item()),
            "constrained_fraction": float(((u - tu).abs().amax(dim=1) > 1e-6).double().mean().item())}


def backup_cbf_leg(dev, B=4096, steps=5, seed=0, split=True, fleet=None):
    """Backup-CBF QP on the reference's evade scenario, on states OF ITS CLOSED LOOP: a fleet starts like the
    example (examples/evade/test_evade.py: robot at the hallway entrance, bullet behind it) with staggered start positions and
    bullet offsets, runs the example's loop for 0 .. 560 control steps on the device (the example runs 600: the whole bullet cycle), and the
    timed launches solve the QP at the states it is in then (round 2 drew states uniformly over the hallway: three quarters of those QPs were infeasible and the
    figure was mostly the fallback branch).  Per agent: a 120-state backup rollout with forward-difference sensitivities, <= 120
    rows, exact QP (csrc/backup_cbf.hip).  The solved and the fallback sub-batches are also timed on their own."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    rng = np.random.default_rng(seed)
    ctl = sca.BatchedBackupCBF(io_dtype="f32")
    if fleet is not None:                                              # tools/prof_backup.py: a fleet prepared outside the profiled process
        tX, bx = fleet
    else:
        # the example's start (robot at x = 20, bullet 30 m behind it, examples/evade/test_evade.py) with a little spread; agent i then
        # runs the example's loop for 40 * (i mod 15) control steps, so the fleet samples the whole bullet cycle of the 600-step run
        X = np.column_stack([20.0 + rng.uniform(-2.0, 2.0, B), rng.uniform(-0.5, 0.5, B), np.zeros(B), np.zeros(B)])
        tX = torch.tensor(X, dtype=torch.float32, device=dev)
        bx = torch.tensor(-10.0 + rng.uniform(-3.0, 3.0, B), dtype=torch.float32, device=dev)
        ret = torch.zeros(B, dtype=torch.int32, device=dev); rs = torch.full((B,), -1, dtype=torch.int32, device=dev)
        groups = 15
        done = 0
        for g_ in range(1, groups):                                        # agents sorted by how long they run: the first m continue
            m = (B * (groups - g_)) // groups
            sub = (tX[:m].contiguous(), bx[:m].contiguous(), ret[:m].contiguous(), rs[:m].contiguous())
            ctl.rollout(*sub, 40, step_offset=done)
            tX[:m], bx[:m], ret[:m], rs[:m] = sub
            done += 40
        alive = ret == 0
        tX, bx = tX[alive].contiguous(), bx[alive].contiguous()
    Bn = int(tX.shape[0])
    if steps == 0:
        return tX, bx

    def timed(Xs, bs):
        u, st, using, hmin = ctl.solve(Xs, None, bs)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            u, st, using, hmin = ctl.solve(Xs, None, bs)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps, st, using
    ms, st, using = timed(tX, bx)
    out = {"workload": f"{Bn} DoubleIntegrator2D agents on states of the evade example's closed loop (0 - 560 control steps in), Backup-CBF "
                       "QP: 120 backup states, forward-difference sensitivities, <= 120 rows, 2 inputs",
           "value": Bn / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32", "agents": Bn,
           "qp_solved_fraction": float((st == 0).double().mean().item()),
           "qp_no_rows_fraction": float((st == -1).double().mean().item()),
           "qp_infeasible_fraction": float((st == 1).double().mean().item()),
           "using_backup_fraction": float((using != 0).double().mean().item())}
    for name, sel in (("solved", st == 0), ("fallback", st == 1)):
        n = int(sel.sum().item())
        if split and n >= 64:
            ms_s, _, _ = timed(tX[sel].contiguous(), bx[sel].contiguous())
            out[f"{name}_only"] = {"agents": n, "kernel_ms": ms_s, "value": n / (ms_s * 1e-3), "unit": "solves/s"}
    rl = valu_roofline("backup_sq", "backupcbf_kernel", ms)         # tools/prof_backup.py: the same fleet, full-batch launches only
    if rl is not None:
        out["roofline"] = rl
    return out


def linear_mpc_leg(dev, model, B=4096, K=8, N=10, steps=3, seed=0):
    """MPC-CBF on the reference's linear models (BASELINE config 5 names Quad3D): B agents, horizon N,
    K obstacles, one NLP per wavefront (csrc/mpc_lin.hip)."""
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    mk = lambda **kw: sca.BatchedLinearMPCCBF({"model": model}, io_dtype="f32", horizon=N, **kw)   # noqa: E731
    ctl = mk()
    Xn, gn, on = W.linear_mpc_batch(model, B, K, seed=seed)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob = t(Xn), t(gn), t(on)
    up = torch.zeros((B, Xn.shape[1] == 12 and 4 or 2), dtype=torch.float32, device=dev)
    u, st, it = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        u, st, it = ctl.solve(X, up, g, ob)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nxu = "12, 4" if model == "Quad3D" else "2, 2"
    kname = f"mpclin_kernel<{nxu}, {N}, {K}, false, false>" if N == 10 else f"mpclin_kernel<{nxu}, 0, 0, true, false>"
    extra = {}
    if model == "SingleIntegrator2D" and N <= 62 and K <= 16:          # the same batch on kernel 13 (the reference's formulation; on request for this robot)
        msc = sca.BatchedMSMPCCBF({"model": model}, io_dtype="f32", horizon=N, check_circles=False)
        um, sm, im = msc.solve(X, up, g, ob)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            um, sm, im = msc.solve(X, up, g, ob)
        e1.record()
        torch.cuda.synchronize()
        mms = e0.elapsed_time(e1) / steps
        both = (sm == 0) & (st == 0)
        extra["multiple_shooting"] = {"kernel": "mpcdu_ms_kernel<float, 4> (kernel 13; on request: mpc_formulation = 'multiple_shooting')", "value": B / (mms * 1e-3), "kernel_ms": mms,
                                      "optimal_fraction": float((sm == 0).double().mean().item()), "max_ipm_iterations": int(im.max().item()),
                                      "same_u0_where_both_optimal_fraction": float(((um - u).abs().amax(dim=1) <= 1e-4)[both].double().mean().item()) if bool(both.any()) else None}
        rl = valu_roofline("dumssi_sq", "mpcdu_ms_kernel<float, 4,", mms) if (B, K, N, seed) == (4096, 8, 10, 0) else None
        if rl:
            extra["multiple_shooting"]["roofline"] = rl
    return with_roofline({**extra, **budget_note(mk, (X, up, g, ob), steps, ms, st, it),
            "workload": f"{B}-agent batch {model} MPC-CBF, horizon N={N}, {K} obstacles ({N * up.shape[1]} variables)",
            "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32",
            "optimal_fraction": float((st == 0).double().mean().item()),
            "infeasible_fraction": float((st == 1).double().mean().item()),
            "mean_ipm_iterations": float(it.double().mean().item())}, kname)


def gn_mpc_leg(dev, model, B=4096, K=8, N=10, steps=3, seed=0):
    """MPC-CBF for DoubleIntegrator2D / Quad2D / KinematicBicycle2D (csrc/mpc_gn.hip): the barrier steps the state
    with the robot's own step(), one NLP per wavefront."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    mk = lambda **kw: sca.BatchedGnMPCCBF({"model": model}, io_dtype="f32", horizon=N, **kw)   # noqa: E731
    ctl = mk()
    fam = {v: k for k, v in W.MPC_FAMILIES.items()}[model]
    Xn, up0, gn, on = W.mpc_family_batch(fam, B, K, seed=seed)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob, up = t(Xn), t(gn), t(on), t(up0)
    u, st, it = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        u, st, it = ctl.solve(X, up, g, ob)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    mid = {"DoubleIntegrator2D": 5, "Quad2D": 6, "KinematicBicycle2D": 1, "KinematicBicycle2D_C3BF": 2, "KinematicBicycle2D_DPCBF": 3}[model]
    extra = {}
    if model in ("DoubleIntegrator2D", "KinematicBicycle2D") and N <= 62 and K <= 16:
        # the same batch in the reference's own formulation: kernel 13 instantiated for this robot (multiple shooting under IPOPT's algorithm)
        msc = sca.BatchedMSMPCCBF({"model": model}, io_dtype="f32", horizon=N, check_circles=False)
        um, sm, im = msc.solve(X, up, g, ob)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            um, sm, im = msc.solve(X, up, g, ob)
        e1.record()
        torch.cuda.synchronize()
        mms = e0.elapsed_time(e1) / steps
        both = (sm == 0) & (st == 0)
        mi = 1 if model == "DoubleIntegrator2D" else 2
        ms_d = {"kernel": f"mpcdu_ms_kernel<float, {mi}> (kernel 13: multiple shooting, IPOPT's filter interior point; " +
                          ("the drop-in's default for this robot)" if mi == 1 else "on request: mpc_formulation = 'multiple_shooting')"),
                "value": B / (mms * 1e-3), "kernel_ms": mms, "optimal_fraction": float((sm == 0).double().mean().item()),
                "infeasible_fraction": float((sm == 1).double().mean().item()), "inaccurate_fraction": float((sm == 2).double().mean().item()),
                "mean_ipm_iterations": float(im.double().mean().item()), "max_ipm_iterations": int(im.max().item()),
                "same_status_fraction": float((sm == st).double().mean().item()),
                "same_u0_where_both_optimal_fraction": float(((um - u).abs().amax(dim=1) <= 1e-4)[both].double().mean().item()) if bool(both.any()) else None}
        rl = valu_roofline("dumsdi_sq" if mi == 1 else "dumskb_sq", f"mpcdu_ms_kernel<float, {mi},", mms) if (B, K, N, seed) == (4096, 8, 10, 0) else None
        if rl:
            ms_d["roofline"] = rl
        if int(im.max().item()) > 100 and not NO_LIMIT100:                  # (the bicycle: solves that cycle around the kink of robot.step's speed clip run to the budget)
            m100 = sca.BatchedMSMPCCBF({"model": model}, io_dtype="f32", horizon=N, check_circles=False, max_iter=100)
            o100 = m100.solve(X, up, g, ob)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                o100 = m100.solve(X, up, g, ob)
            e1.record()
            torch.cuda.synchronize()
            ms_d["beyond_100_iterations"] = int((im > 100).sum().item())
            ms_d["one_launch_limit_100"] = {"kernel_ms": e0.elapsed_time(e1) / steps, "optimal_fraction": float((o100[1] == 0).double().mean().item())}
        extra["multiple_shooting"] = ms_d
    return with_roofline({**extra, **budget_note(mk, (X, up, g, ob), steps, ms, st, it),
            "workload": f"{B}-agent batch {model} MPC-CBF, horizon N={N}, {K} obstacles",
            "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32",
            "optimal_fraction": float((st == 0).double().mean().item()),
            "infeasible_fraction": float((st == 1).double().mean().item()),
            "mean_ipm_iterations": float(it.double().mean().item())}, f"mpcgn_kernel<{mid}, {N if N == 10 else 0}, false>")


def unicycle_mpc_leg(dev, B=4096, K=8, N=10, steps=3, seed=0):
    """MPC-CBF for Unicycle2D (three states, inputs (v, omega), one-step rows): kernel 13's instantiation for this robot (the drop-in's default
    since the end of round 6) with the condensed kernel 3 on the same batch beside it."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    Xn, up0, gn, on = W.mpc_family_batch("uni", B, K, seed=seed)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)     # noqa: E731
    X, g, ob, up = t(Xn), t(gn), t(on), t(up0)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)

    def timed(ctl):
        out = ctl.solve(X, up, g, ob)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            out = ctl.solve(X, up, g, ob)
        e1.record()
        torch.cuda.synchronize()
        return out, e0.elapsed_time(e1) / steps

    (u, st, it), ms = timed(sca.BatchedMSMPCCBF({"model": "Unicycle2D"}, io_dtype="f32", horizon=N, check_circles=False))
    (uc, sc_, ic), msc = timed(sca.BatchedMPCCBF({"model": "Unicycle2D"}, io_dtype="f32", horizon=N))
    both = (st == 0) & (sc_ == 0)
    res = {"workload": f"{B}-agent batch Unicycle2D MPC-CBF, horizon N={N}, {K} obstacles", "kernel": "mpcdu_ms_kernel<float, 3> (kernel 13)",
           "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32",
           "optimal_fraction": float((st == 0).double().mean().item()), "infeasible_fraction": float((st == 1).double().mean().item()),
           "inaccurate_fraction": float((st == 2).double().mean().item()),
           "mean_ipm_iterations": float(it.double().mean().item()), "max_ipm_iterations": int(it.max().item()),
           "condensed": {"kernel": "mpccbf_uni_kernel (kernel 3)", "value": B / (msc * 1e-3), "kernel_ms": msc, "optimal_fraction": float((sc_ == 0).double().mean().item()),
                         "inaccurate_fraction": float((sc_ == 2).double().mean().item()), "max_ipm_iterations": int(ic.max().item()),
                         "same_u0_where_both_optimal_fraction": float(((u - uc).abs().amax(dim=1) <= 1e-4)[both].double().mean().item()) if bool(both.any()) else None}}
    rl = valu_roofline("dumsuni_sq", "mpcdu_ms_kernel<float, 3,", ms) if (B, K, N, seed) == (4096, 8, 10, 0) else None
    if rl:
        res["roofline"] = rl
    return res


def bicycle_loop_states(dev, model, B=4096, T=160, every=10, seed=0):
    """MPC-CBF of the collision-cone bicycles on states OF THEIR CLOSED LOOP (round 3 timed batches drawn uniformly, half of them inside
    collision cones: half the "solves" were restorations).  The fleet flies dynamic_env/main.py's scene (:241-268: start (1, 7.5),
    heading 0, 1 m/s; goal (20, 7.5); eight discs of radius 0.5 moving at (-0.5, +-0.5) m/s, stepped like step_dyn_obs, :54-58) with
    start positions, headings and speeds spread a little; agent i is sampled after `every` * (i mod 16) control steps, so the timed batch
    covers the whole approach.  Returns X[Bn,4], u_prev[Bn,2], goal[Bn,2], obs[Bn,8,7] (f32, on the device), G, every."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    rng = np.random.default_rng(seed)
    dt = 0.05
    table = np.array([[8.0, 9.0], [10.0, 4.0], [12.0, 5.0], [14.0, 9.0], [16.0, 6.0], [18.0, 14.0], [20.0, 4.0], [22.0, 12.0]])
    obs = np.zeros((8, 7)); obs[:, :2] = table; obs[:, 2] = 0.5
    obs[:, 3] = -0.5; obs[:, 4] = np.where(np.arange(8) % 2 == 0, 0.5, -0.5)
    X0 = np.column_stack([1.0 + rng.uniform(-0.5, 0.5, B), 7.5 + rng.uniform(-1.5, 1.5, B), rng.uniform(-0.3, 0.3, B), 1.0 + rng.uniform(0.0, 1.0, B)])
    spec = {"model": model, "a_max": 5.0, "radius": 0.3, "num_constraints": 8}
    ctl = sca.BatchedTrackingController(X0, spec, controller_type={"pos": "mpc_cbf"}, obs=obs, io_dtype="f32", device=str(dev))
    ctl.mpc.max_iter, ctl.mpc.iter_slices, ctl.mpc.classify_first = 100, (), False      # preparing the fleet: the round-3 limit is enough
    ctl.set_waypoints(np.array([[1.0, 7.5], [20.0, 7.5]]))
    G = 16
    grp = torch.arange(B, device=dev) % G
    Xs = torch.zeros((B, 4), dtype=torch.float32, device=dev); Us = torch.zeros((B, 2), dtype=torch.float32, device=dev)
    Os = torch.zeros((B, 8, 7), dtype=torch.float32, device=dev); live = torch.zeros(B, dtype=torch.bool, device=dev)
    for step in range(min(T, every * (G - 1)) + 1):
        if step % every == 0:
            m = grp == step // every
            Xs[m] = ctl.X[m]; Us[m] = ctl.u_prev[m]; live[m] = ctl.ret[m] == 0
            Os[m] = torch.tensor(obs, dtype=torch.float32, device=dev)
        ctl.set_obstacles(obs)
        ctl.control_step(1)
        obs[:, 0] += obs[:, 3] * dt; obs[:, 1] += obs[:, 4] * dt
    X, up, ob = Xs[live].contiguous(), Us[live].contiguous(), Os[live].contiguous()
    Bn = int(X.shape[0])
    g = torch.tensor([[20.0, 7.5]], dtype=torch.float32, device=dev).repeat(Bn, 1).contiguous()
    return X, up, g, ob, G, every


def bicycle_loop_leg(dev, model, B=4096, T=160, every=10, steps=3, seed=0):
    """MPC-CBF of the collision-cone bicycles on states OF THEIR CLOSED LOOP (bicycle_loop_states; round 3 timed batches drawn
    uniformly, half of them inside collision cones: half the "solves" were restorations).  Reported: the batch rate and the rate of
    the problems that end optimal."""
    import torch
    import safe_control_amd as sca
    X, up, g, ob, G, every = bicycle_loop_states(dev, model, B, T, every, seed)
    Bn = int(X.shape[0])
    mk = lambda **kw: sca.BatchedGnMPCCBF({"model": model, "a_max": 5.0, "radius": 0.3}, io_dtype="f32", horizon=10, **kw)   # noqa: E731

    def timed(c, args):
        out = c.solve(*args)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            out = c.solve(*args)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps, out[1], out[2]
    ms, st, it = timed(mk(), (X, up, g, ob))
    res = {**budget_note(mk, (X, up, g, ob), steps, ms, st, it),
           "workload": f"{Bn} {model} agents on states of the dynamic_env closed loop (0 - {every * (G - 1)} control steps in, eight moving discs), MPC-CBF N=10",
           "value": Bn / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32", "agents": Bn,
           "optimal_fraction": float((st == 0).double().mean().item()), "infeasible_fraction": float((st == 1).double().mean().item()),
           "mean_ipm_iterations": float(it.double().mean().item())}
    opt = st == 0
    if int(opt.sum()) >= 64:
        ms_o, _, _ = timed(mk(), (X[opt].contiguous(), up[opt].contiguous(), g[opt].contiguous(), ob[opt].contiguous()))
        res["optimal_only_value"] = int(opt.sum()) / (ms_o * 1e-3)
        res["optimal_only_kernel_ms"] = ms_o
    return res


def vtol_mpc_leg(dev, B=4096, K=8, steps=2, seed=0):
    """MPC-CBF for VTOL2D (csrc/mpc_vtol_wave.hip): N = 30, 4 inputs, one NLP per wavefront, one stage per lane, stage-wise
    Riccati Newton steps, rows in registers, everything else in LDS (no workspace)."""
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    mk = lambda **kw: sca.BatchedVtolMPCCBF(io_dtype="f32", **kw)   # noqa: E731
    ctl = mk()
    Xn, up0, gn, on = W.mpc_family_batch("vtol", B, K, seed=seed)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob, up = t(Xn), t(gn), t(on), t(up0)
    u, st, it = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        u, st, it = ctl.solve(X, up, g, ob)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return with_roofline({**budget_note(mk, (X, up, g, ob), steps, ms, st, it),
            "workload": f"{B}-aircraft batch VTOL2D MPC-CBF, horizon N=30, {K} obstacles (120 variables, 630 rows per NLP)",
            "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32",
            "optimal_fraction": float((st == 0).double().mean().item()),
            "infeasible_fraction": float((st == 1).double().mean().item()),
            "mean_ipm_iterations": float(it.double().mean().item()),
            "lds_KB_per_problem": 39.3, "problems_per_CU": 4}, "mpcvtol_wave_kernel<float, 8")   # (<float, 8> in the profiles of rounds 3; <float, 8, false> since the OD flag)


def vtol_ms_mpc_leg(dev, B=4096, K=8, steps=3, seed=0):
    """MPC-CBF for VTOL2D AS DO-MPC POSES IT (round 5, csrc/mpc_vtol_ms.hip, DESIGN.md kernel 12): multiple shooting (states of every stage are
    variables, dynamics as equality rows, x_k = x0 start: position_control/mpc_cbf.py:162-174,366-369) under IPOPT's filter line-search
    interior point with IPOPT's option defaults (tol 1e-8, max_iter 3000), one NLP per wavefront, one stage per lane, Riccati recursion with
    defects on the f64 matrix cores, IPOPT's restoration phase inside the kernel (no solve of this batch enters it).  One launch: the
    longest solve of the batch is ~100 iterations, no continuation needed.  Same batch as vtol_mpc_cbf."""
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    ctl = sca.BatchedVtolMSMPCCBF(io_dtype="f32")
    Xn, up0, gn, on = W.mpc_family_batch("vtol", B, K, seed=seed)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob, up = t(Xn), t(gn), t(on), t(up0)
    u, st, it = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        u, st, it = ctl.solve(X, up, g, ob)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    res = {"workload": f"{B}-aircraft batch VTOL2D MPC-CBF, multiple shooting (306 variables, 186 equality rows, {30 * K} inequality rows per NLP), horizon N=30, {K} obstacles",
           "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32", "budget": 3000, "launches": 1,
           "optimal_fraction": float((st == 0).double().mean().item()), "inaccurate_fraction": float((st == 2).double().mean().item()),
           "restoration_fallback": int(ctl.n_fallback), "mean_ipm_iterations": float(it.double().mean().item()), "max_ipm_iterations": int(it.max().item()),
           "lds_KB_per_problem": 40.0, "problems_per_CU": 4}
    # the work without the tail: the batch filled with copies of the median problem
    try:
        opt = torch.nonzero(st == 0).flatten()
        med = opt[torch.argsort(it[opt])[opt.numel() // 2]]
        rep = lambda a: a[med:med + 1].repeat(B, *([1] * (a.dim() - 1))).contiguous()
        args = (rep(X), rep(up), rep(g), rep(ob))
        ctl.solve(*args); torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            o2 = ctl.solve(*args)
        e1.record(); torch.cuda.synchronize()
        res["uniform_batch"] = {"kernel_ms": e0.elapsed_time(e1) / steps, "iterations": int(o2[2][0].item())}
    except Exception as e:
        res["uniform_batch"] = {"error": repr(e)[:100]}
    return with_roofline(res, "mpcvtol_ms_kernel<float, 8, false>", launches=1)


def vtol_ms_closed_loop_leg(dev):
    """The reference's own VTOL2D demo (examples/test_vtol.py:12-92) through the drop-in loop with the default position controller of the model
    (the multiple-shooting kernel, restoration phase included): one aircraft, control steps until the loop returns."""
    import time
    import numpy as np
    import torch
    import safe_control_amd as sca
    obs = np.array([[67.0, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[73.0, float(z), 0.5] for z in range(1, 16)] + [[60.0, 12.0, 1.5]])
    obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
    spec = {"model": "VTOL2D", "radius": 0.6, "v_max": 20.0, "reached_threshold": 1.0, "num_constraints": 10}
    ctl = sca.BatchedTrackingController(np.array([[2.0, 10.0, 0.0, 20.0, 0.0, 0.0]]), spec, obs=obs7, device=str(dev))
    ctl.set_waypoints(np.array([[2.0, 10.0], [70.0, 10.0], [70.0, 0.5]]))
    torch.cuda.synchronize()
    t0 = time.time()
    ret, n, nopt = 0, 0, 0
    for n in range(1, 401):
        ret = int(ctl.control_step(1)[0].item())
        nopt += int(ctl.mpc_status[0].item() == 0)
        if ret != 0:
            break
    return {"workload": "examples/test_vtol.py scene, one VTOL2D aircraft, MPC-CBF N=30, 10 obstacle slots, closed loop to the landing waypoint",
            "return_code": ret, "landed": ret == -1, "control_steps": n, "optimal_solves": nopt, "ms_per_control_step": 1e3 * (time.time() - t0) / n}


def vtol_fleet_closed_loop_leg(dev, B=256, seed=0):
    """The same scene flown by B aircraft at once (starts spread over 10 m of approach and 1 m of altitude, 18 - 20 m/s; waypoints (70, 10) then
    (70, 0.5)): one control_step of the batched loop = select the nearest obstacles, ONE launch of the multiple-shooting kernel for every aircraft
    that is tracking, apply.  Runs until every aircraft has returned (landed = -1, collision = -2) or 450 steps; once with the reference solver's
    budget (3000 iterations: a step lasts as long as its hardest NLP -- the first ~50 steps hold infeasible approaches) and once with
    robot_spec['mpc_max_iter'] = 200."""
    import time
    import numpy as np
    import torch
    import safe_control_amd as sca
    obs = np.array([[67.0, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[73.0, float(z), 0.5] for z in range(1, 16)] + [[60.0, 12.0, 1.5]])
    obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
    rng = np.random.default_rng(seed)
    X0 = np.zeros((B, 6))
    X0[:, 0] = 2.0 + 10.0 * rng.uniform(size=B); X0[:, 1] = 10.0 + rng.uniform(-0.5, 0.5, B); X0[:, 3] = rng.uniform(18.0, 20.0, B)
    X0[0] = [2.0, 10.0, 0.0, 20.0, 0.0, 0.0]

    def fly(extra):
        spec = dict({"model": "VTOL2D", "radius": 0.6, "v_max": 20.0, "reached_threshold": 1.0, "num_constraints": 10}, **extra)
        ctl = sca.BatchedTrackingController(X0, spec, obs=obs7, device=str(dev))
        ctl.set_waypoints(np.array([[70.0, 10.0], [70.0, 0.5]]))            # (the demo's first waypoint is its start: not a target for the others)
        torch.cuda.synchronize()
        t0 = time.time()
        done = torch.zeros(B, dtype=torch.int32, device=dev)
        n, t_late, n_late = 0, 0.0, 0
        for n in range(1, 451):
            ts = time.time()
            ret = ctl.control_step(1)
            done = torch.where((done == 0) & (ret != 0), ret.to(torch.int32), done)
            fin = bool((done != 0).all())
            if n > 100:
                t_late += time.time() - ts; n_late += 1
            if fin:
                break
        dt = time.time() - t0
        return {"control_steps": n, "landed_fraction": float((done == -1).double().mean().item()), "lost_fraction": float((done == -2).double().mean().item()),
                "ms_per_control_step": 1e3 * dt / n, "ms_per_control_step_after_step_100": 1e3 * t_late / max(1, n_late), "agent_steps_per_s": B * n / dt}

    res = {"workload": f"examples/test_vtol.py scene, {B} VTOL2D aircraft at once from perturbed starts, MPC-CBF N=30, 10 obstacle slots, closed loop until every aircraft has returned",
           "aircraft": B}
    res.update(fly({}))
    res["with_mpc_max_iter_200"] = fly({"mpc_max_iter": 200})
    return res


def od_vtol_mpc_leg(dev, B=4096, K=8, steps=2, seed=0):
    """Optimal-decay MPC-CBF for VTOL2D (the last model of the reference class's accept list): the vtol batch with a disc
    on every other aircraft's flight path 10 - 30 m ahead, so that decay variables leave their reference.  One launch per solve."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    ctl = sca.BatchedOptimalDecayVtolMPCCBF(io_dtype="f32")
    Xn, up0, gn, on = W.mpc_family_batch("vtol", B, K, seed=seed)
    on = on.copy()
    rng = np.random.default_rng(seed + 100)
    r = rng.uniform(0.8, 1.6, B); d = 10.0 + 20.0 * rng.uniform(size=B); off = rng.uniform(-1.0, 1.0, B)
    on[::2, 0, 0], on[::2, 0, 1], on[::2, 0, 2] = (Xn[:, 0] + d + r)[::2], (Xn[:, 1] + off)[::2], r[::2]
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob, up = t(Xn), t(gn), t(on), t(up0)
    u, rho, st, it = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        u, rho, st, it = ctl.solve(X, up, g, ob)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    lim = {}
    if not NO_LIMIT100:                                              # the same batch stopped at 100 iterations (one solve crawls to the budget)
        c100 = sca.BatchedOptimalDecayVtolMPCCBF(io_dtype="f32", max_iter=100)
        o100 = c100.solve(X, up, g, ob)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            o100 = c100.solve(X, up, g, ob)
        e1.record()
        torch.cuda.synchronize()
        lim = {"one_launch_limit_100": {"kernel_ms": e0.elapsed_time(e1) / steps, "optimal_fraction": float((o100[2] == 0).double().mean().item())},
               "beyond_100_iterations": int((it > 100).sum().item()), "inaccurate_fraction": float((st == 2).double().mean().item())}
    return with_roofline({**lim, "workload": f"{B}-aircraft batch VTOL2D optimal-decay MPC-CBF, horizon N=30, {K} obstacles, a disc ahead of every other aircraft",
            "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32",
            "optimal_fraction": float((st == 0).double().mean().item()),
            "infeasible_fraction": float((st == 1).double().mean().item()),
            "mean_ipm_iterations": float(it.double().mean().item()), "max_ipm_iterations": int(it.max().item()),
            "decay_moved_fraction": float(((rho - 1.0).abs().max(dim=1).values > 1e-3).double().mean().item())},
            "mpcvtol_wave_kernel<float, 8, true>", launches=1)


def od_vtol_ms_mpc_leg(dev, B=4096, K=8, steps=2, seed=0):
    """Optimal-decay MPC-CBF for VTOL2D in the multiple-shooting form (round 5: the OD instantiation of csrc/mpc_vtol_ms.hip; the decay rates are
    two more inputs of a stage, as in the reference, optimal_decay_mpc_cbf.py:123-124).  Same batch as od_vtol_mpc_cbf; problems that would
    enter IPOPT's restoration phase run it inside the kernel (elastic variables on the CBF rows)."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    from safe_control_amd import workloads as W
    ctl = sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f32")
    Xn, up0, gn, on = W.mpc_family_batch("vtol", B, K, seed=seed)
    on = on.copy()
    rng = np.random.default_rng(seed + 100)
    r = rng.uniform(0.8, 1.6, B); d = 10.0 + 20.0 * rng.uniform(size=B); off = rng.uniform(-1.0, 1.0, B)
    on[::2, 0, 0], on[::2, 0, 1], on[::2, 0, 2] = (Xn[:, 0] + d + r)[::2], (Xn[:, 1] + off)[::2], r[::2]
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob, up = t(Xn), t(gn), t(on), t(up0)
    u, rho, st, it = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        u, rho, st, it = ctl.solve(X, up, g, ob)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return with_roofline({"workload": f"{B}-aircraft batch VTOL2D optimal-decay MPC-CBF, multiple shooting, horizon N=30, {K} obstacles, a disc ahead of every other aircraft",
            "value": B / (ms * 1e-3), "unit": "solves/s", "kernel_ms": ms, "dtype": "f64", "storage": "f32", "budget": 3000,
            "optimal_fraction": float((st == 0).double().mean().item()), "infeasible_fraction": float((st == 1).double().mean().item()),
            "inaccurate_fraction": float((st == 2).double().mean().item()), "restoration_fallback": int(ctl.n_fallback),
            "mean_ipm_iterations": float(it.double().mean().item()), "max_ipm_iterations": int(it.max().item()),
            "decay_moved_fraction": float(((rho - 1.0).abs().max(dim=1).values > 1e-3).double().mean().item())},
            "mpcvtol_ms_kernel<float, 8, true>", launches=1)


def manip_closed_loop_leg(dev, B=4096, T=100, seed=0):
    """Fused closed loop for B arms (csrc/manip_cbf_qp.hip: manip_rollout_kernel): T control steps in one launch, four shared
    obstacles around the workspace, two waypoints per arm."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    rng = np.random.default_rng(seed)
    base = np.array([5.0, 3.5])
    obs = np.array([[6.6, 5.2, 0.3], [3.4, 5.4, 0.3], [7.4, 2.2, 0.3], [3.0, 1.8, 0.35]])
    q0 = rng.uniform(-1.2, 1.2, (B, 3))
    ang, rad = rng.uniform(-np.pi, np.pi, (B, 2)), rng.uniform(1.6, 3.0, (B, 2))
    wl = [np.stack([base[0] + rad[i] * np.cos(ang[i]), base[1] + rad[i] * np.sin(ang[i])], axis=1) for i in range(B)]
    spec = {"model": "Manipulator2D", "w_max": 2.0, "Kp": 5.0, "radius": 0.25, "reached_threshold": 0.4}
    ctl = sca.BatchedManipulatorTracking(q0, dict(spec), base_pos=base, obs=obs, io_dtype="f32", device=dev)
    ctl.set_waypoints(wl)
    ctl.control_step(1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    ret = ctl.control_step(T)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    return {"workload": f"{B} Manipulator2D arms x {T} closed-loop control steps, 4 shared obstacles (100 CBF rows + 6 box rows), one launch",
            "kernel_ms": ms, "agent_steps_per_s": B * T / (ms * 1e-3), "finished": int((ret == -1).sum().item()),
            "failed": int((ret == -2).sum().item()), "running": int((ret == 0).sum().item()), "dtype": "f64", "storage": "f32"}


def closed_loop_leg(dev, B=4096, T=200, seed=0):
    """BASELINE config 2, closed-loop variant: B DynamicUnicycle2D agents track waypoints through the 14-circle
    scene of examples/test_tracking.py for T control steps (selection + nominal input + CBF-QP + step + collision
    check per step) in ONE launch of the fused rollout kernel."""
    import numpy as np
    import torch
    import safe_control_amd as sca
    obs = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0], [7.0, 7.0, 3.0],
                    [4.0, 3.5, 1.5], [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6], [11.0, 5.0, 0.8],
                    [13.5, 11.0, 0.6], [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
    rng = np.random.default_rng(seed)
    X0 = np.zeros((0, 4))
    while len(X0) < B:                                                    # collision-free start poses
        p = rng.uniform(0.5, 13.5, (2 * B, 2))
        ok = np.min(np.hypot(obs[None, :, 0] - p[:, None, 0], obs[None, :, 1] - p[:, None, 1]) - obs[None, :, 2], axis=1) > 0.6
        p = p[ok]
        X0 = np.vstack([X0, np.hstack([p, rng.uniform(-np.pi, np.pi, (len(p), 1)), rng.uniform(0, 1, (len(p), 1))])])
    X0 = X0[:B]
    wps = [rng.uniform(1, 13, (3, 2)) for _ in range(B)]
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    ctl = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f32", device=str(dev))
    ctl.set_waypoints(wps)
    ctl.control_step(1)                                                   # warm-up launch
    torch.cuda.synchronize()
    # one launch per run: the event bracket also holds the host's launch latency, so the run is repeated and the fastest kept
    ms = float("inf")
    for _ in range(3):
        ctl = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f32", device=str(dev))
        ctl.set_waypoints(wps)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ret = ctl.control_step(T)
        e1.record()
        torch.cuda.synchronize()
        ms = min(ms, e0.elapsed_time(e1))
    ret = ret.cpu().numpy(); rs = ctl.ret_step.cpu().numpy()
    steps_run = np.where(ret == 0, T, rs + 1).sum()
    return {"workload": f"{B} DynamicUnicycle2D agents x {T} closed-loop control steps, 14 shared obstacles, num_constraints 10, one launch",
            "kernel_ms": ms, "agent_steps_per_s": float(steps_run) / (ms * 1e-3), "agent_steps": int(steps_run),
            "finished": int((ret == -1).sum()), "failed": int((ret == -2).sum()), "running": int((ret == 0).sum()),
            "dtype": "f64", "storage": "f32"}


def self_launch(a):
    """``python bench.py --gpus N`` with N > 1 and no launcher environment: this process starts the N ranks itself, as a CHILD
    (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>``), before anything here has
    imported torch or touched the GPU; it relays the child's output (rank 0's JSON line) and exit code, and fails if no line came
    back.  Under torchrun (WORLD_SIZE set) it does nothing.  Returns the exit code, or None when there is nothing to launch."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import socket
    import subprocess
    with socket.socket() as sk:                                    # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    got_line = False
    for line in proc.stdout:
        got_line = got_line or line.startswith("{")
        print(line, end="", flush=True)
    rc = proc.wait()
    if rc == 0 and not got_line:
        print(f"bench.py: the {a.gpus} ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
LINE_LIMIT = 7000                # the driver keeps the last 8 KB of stdout: the whole line has to fit


def compact_leg(v):
    """What the one JSON line keeps of a leg: its rate, launch time, solved fraction and roofline fraction.  The full dict goes to
    gpurun_out/bench_legs.json."""
    if isinstance(v, list):
        return [compact_leg(x) for x in v]
    if not isinstance(v, dict):
        return v
    def sig(x):                                                          # four significant digits; rates as integers (two bytes each on a 7 KB line)
        if not isinstance(x, float):
            return x
        y = float(f"{x:.4g}")
        return int(y) if abs(y) >= 1e4 and y == int(y) else y
    keep = {}
    for k in ("value", "kernel_ms", "ms_per_step", "ms_per_control_step", "us_per_step", "agent_steps_per_s", "solves_per_s", "optimal_fraction",
              "max_ipm_iterations", "agents", "GBs", "frac_of_peak", "error", "optimal_only_value",
              "inaccurate_fraction", "landed", "return_code", "control_steps", "restoration_fallback", "landed_fraction", "lost_fraction", "aircraft",
              "all_gather_bytes_per_step", "scaling"):
        if k in v and not (k in ("inaccurate_fraction", "restoration_fallback") and v[k] == 0.0) and v[k] is not None:      # (no inaccurate solves: not worth 28 bytes of the line)
            keep[{"max_ipm_iterations": "max_iter", "inaccurate_fraction": "inaccurate"}.get(k, k)] = sig(v[k])      # (the line is 7 KB for ~35 legs: two key names shortened)
    if isinstance(v.get("one_launch_limit_100"), dict) and v.get("beyond_100_iterations"):      # (only where the budget beyond 100 iterations is used)
        keep["limit_100_ms"] = sig(v["one_launch_limit_100"]["kernel_ms"])
    if isinstance(v.get("multiple_shooting"), dict):                     # the same batch on kernel 13 (the reference's own formulation)
        keep["ms"] = compact_leg(v["multiple_shooting"])
    if isinstance(v.get("condensed"), dict) and "kernel_ms" in v["condensed"]:      # ... or, where kernel 13 is the leg, the condensed kernel beside it
        keep["condensed_ms"] = sig(v["condensed"]["kernel_ms"])
    rl = v.get("roofline")
    if isinstance(rl, dict):
        # (no "stale" per leg: emit() counts the legs whose counter profile is not from this tree's kernels under `stale_rooflines`; bench_legs.json has the flags)
        keep["roofline"] = {k: ("valu" if rl[k] == "valu_issue" else sig(rl[k])) for k in ("bound", "frac") if k in rl}
        if isinstance(rl.get("work_level"), dict):
            keep["roofline"]["work_frac"] = sig(rl["work_level"]["frac"])
    return keep


def emit(d, ws):
    """rank 0's ONE JSON line; ``ranks_seen`` is the world size the process group reported, not the --gpus argument.  The contract
    fields travel in full; every other leg is cut down to compact_leg() and its full dict is written to gpurun_out/bench_legs.json
    (the driver's record keeps 8 KB of stdout, round 3's 14 KB line lost config 3 there).  BASELINE configs[2] (``mpc_cbf``) is
    printed LAST, with its roofline and its own cpu_baseline, so that it is in whatever tail of the line survives; its numbers are
    also copied into ``config`` (a contract field)."""
    d["ranks_seen"] = ws
    full = dict(d)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_legs.json"), "w") as f:
            json.dump(full, f, indent=1)
    except OSError:
        pass
    line = {k: d[k] for k in CONTRACT_KEYS if k in d}
    if isinstance(line.get("cpu_baseline"), dict):                       # (the full dict is in bench_legs.json)
        line["cpu_baseline"] = {k: (float(f"{v_:.6g}") if isinstance(v_, float) else v_) for k, v_ in line["cpu_baseline"].items()
                                if k in ("value", "unit", "cores", "kind", "sample", "one_core_value", "python_per_agent_loop_value")}
    line["ranks_seen"] = ws
    mpc = d.get("mpc_cbf") or d.get("mpc")
    for k, v in d.items():
        if k in line or k in ("mpc_cbf", "mpc"):
            continue
        line[k] = compact_leg(v)
    if isinstance(mpc, dict):
        m = compact_leg(mpc)
        m.pop("condensed_ms", None)                                      # (the condensed kernel's numbers follow in full)
        m["workload"] = "BASELINE configs[2]: 4096 x DynamicUnicycle2D MPC-CBF N=10 K=8" if "configs[2]" in str(mpc.get("workload")) else mpc.get("workload")
        if isinstance(mpc.get("roofline"), dict):
            m["roofline"] = {k: (float(f"{v_:.5g}") if isinstance(v_, float) else v_) for k, v_ in
                             ((k, mpc["roofline"].get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "stale") if k in mpc["roofline"])}
            if isinstance(mpc["roofline"].get("work_level"), dict):
                m["roofline"]["work_frac"] = mpc["roofline"]["work_level"]["frac"]
        if isinstance(mpc.get("condensed"), dict):                     # the formulation of rounds 1 - 5 on the same batch (kernel 3), and how the two answers compare
            c = mpc["condensed"]
            m["infeasible_fraction"] = mpc.get("infeasible_fraction")
            r4 = lambda x: float(f"{x:.4g}") if isinstance(x, float) else x     # noqa: E731
            m["condensed"] = {**compact_leg(c), "same_status": r4(c.get("same_status_fraction")), "same_u0_both_optimal": r4(c.get("same_u0_where_both_optimal_fraction")),
                              "u0_diff_gt_1e-3_neither_optimal": r4(c.get("u0_differs_by_more_than_1e-3_where_neither_is_optimal_fraction"))}
            m["infeasible_fraction"] = r4(m["infeasible_fraction"])
        if isinstance(mpc.get("cpu_baseline"), dict):
            m["cpu_baseline"] = {k: (float(f"{v_:.5g}") if isinstance(v_, float) else v_) for k, v_ in
                                 ((k, mpc["cpu_baseline"].get(k)) for k in ("value", "unit", "cores", "kind", "sample", "one_core_value", "python_oracle_value"))}
        if isinstance(line.get("config"), dict) and "mpc" not in d:
            line["config"]["mpc_cbf_configs2"] = {"solves_per_s": m.get("value"), "kernel_ms": m.get("kernel_ms"),
                                                  "roofline_frac": (m.get("roofline") or {}).get("frac"),
                                                  "optimal_fraction": m.get("optimal_fraction")}
        line["mpc_cbf" if "mpc_cbf" in d else "mpc"] = m
    line["full_legs"] = "gpurun_out/bench_legs.json"

    def stale_in(v):
        return isinstance(v, dict) and (bool(v.get("stale")) or any(stale_in(x) for x in v.values()))
    # legs whose VALU-issue roofline uses a counter profile of OTHER kernel sources: [] = none; more than four: their number (a full list would push legs off the line)
    stale = sorted(k for k, v in d.items() if stale_in(v))
    line["stale_rooflines"] = stale if len(stale) <= 4 else len(stale)
    s = json.dumps(line)
    if len(s) > LINE_LIMIT:                                             # never the contract fields nor configs[2]: drop the largest extras
        extras = sorted((k for k in line if k not in CONTRACT_KEYS and k not in ("mpc_cbf", "mpc", "ranks_seen", "full_legs")),
                        key=lambda k: -len(json.dumps(line[k])))
        for k in extras:
            line[k] = {"see": "bench_legs.json"}
            s = json.dumps(line)
            if len(s) <= LINE_LIMIT:
                break
    # configs[2] last
    for k in ("mpc_cbf", "mpc"):
        if k in line:
            line[k] = line.pop(k)
    print(json.dumps(line), flush=True)


def main():
    a = parse()
    global NO_LIMIT100
    NO_LIMIT100 = bool(a.no_limit100)
    rc = self_launch(a)
    if rc is not None:
        sys.exit(rc)
    import numpy as np
    import torch
    import torch.distributed as dist
    import safe_control_amd as sca
    from safe_control_amd import sharding, workloads as W

    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)          # (several ranks share a GPU only in the 1-GPU flow test below)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RCCL over xGMI (backend "nccl" on ROCm).  SC_BENCH_BACKEND=gloo exists only to exercise the
    # N > 1 control flow on a 1-GPU box (tests/test_bench_flow_gpu.py); it is never the measured setup.
    backend = os.environ.get("SC_BENCH_BACKEND", "nccl")
    if ws > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if ws > 1:
        ws = dist.get_world_size()                                 # what the process group saw, reported as n_gpus / ranks_seen
    if a.gpus != ws and rank == 0:
        print(f"note: --gpus {a.gpus} but the process group has {ws} rank(s); reporting {ws}", file=sys.stderr)

    B, K = a.agents, a.obstacles
    if a.workload == "mpc_cbf":
        if ws > 1:
            dist.barrier()
        r = mpc_leg(dev, B, K, a.horizon, a.steps, a.warmup, seed=rank,
                    cpu_seconds=(6.0 if (ws == 1 and not a.no_cpu_baseline) else 0.0))
        elapsed = sharding.max_over_ranks(B * a.steps / r["value"], device=dev if backend == "nccl" else None)
        if rank == 0:
            emit({"metric": "QP solves/sec (batched agents)", "value": B * ws * a.steps / elapsed,
                              "unit": "solves/s", "n_gpus": ws, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": r["workload"], "agents_per_gpu": B, "obstacles": K,
                                         "horizon": a.horizon, "storage": "f32"},
                              "roofline": r.get("roofline") or {"bound": "hbm", "achieved": r["achieved_GBs"], "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": r["achieved_GBs"] / HBM_PEAK_GBS, "traffic": None,
                                           "kernel": "mpccbf_kernel", "kernel_us": 1e3 * r["kernel_ms"],
                                           "note": "ALU/LDS-bound interior-point iterations; no VALU counter profile committed for this "
                                                   "configuration, HBM fraction reported for completeness"},
                              "cpu_baseline": r.get("cpu_baseline"), "mpc": r}, ws)
        if ws > 1:
            dist.destroy_process_group()
        return
    if a.workload == "hetero_fleet":
        hetero_fleet_workload(a, dev, ws, rank, backend)
        if ws > 1:
            dist.destroy_process_group()
        return
    if a.workload == "kb_c3bf":
        kb_c3bf_workload(a, dev, ws, rank, backend)
        if ws > 1:
            dist.destroy_process_group()
        return
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    ctl = sca.BatchedCBFQP(dict(spec), dt=0.05, io_dtype=a.io, compute_dtype=a.compute)
    td = ctl.torch_dtype
    es = 4 if a.io == "f32" else 8
    Xn, goal, un, on = W.du_cbfqp_batch(B, K, seed=rank)           # each rank: its own shard
    X = torch.tensor(Xn, dtype=td, device=dev)
    ur = torch.tensor(un, dtype=td, device=dev)
    ob = torch.tensor(on, dtype=td, device=dev)
    out = (torch.empty((B, 2), dtype=td, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
           torch.empty((B, K), dtype=td, device=dev))

    def step():
        ctl.solve(X, ur, ob, out=out)

    for _ in range(max(a.warmup, 1)):
        step()
    torch.cuda.synchronize()

    graph = None
    if not a.eager:
        # the launch-bound inner loop as ONE hipGraph: K kernel nodes on the capture stream
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            step()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(a.steps):
                    step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        graph.replay()                                              # untimed: first replay uploads the graph
        torch.cuda.synchronize()

    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()                                                     # same stream the kernels run on
    if graph is not None:
        graph.replay()
    else:
        for _ in range(a.steps):
            step()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()                                        # this rank's K steps are complete
    if ws > 1:
        dist.barrier()                                              # closing bracket; the job time is the MAX over ranks
    elapsed = sharding.max_over_ranks(t1 - t0, device=dev if backend == "nccl" else None)
    kernel_ms = e0.elapsed_time(e1) / a.steps                       # avg launch duration over the timed region

    st = out[1]
    n_opt = int((st == 0).sum().item())
    # N > 1: the one collective of the design (RCCL all-gather of the agent states, BASELINE configs[3]) is measured in the
    # same run, on every rank, so that a scaling sweep of the default command covers the exchange path as well
    coll = kb_c3bf_workload(a, dev, ws, rank, backend, collect=True, steps_override=20) if ws > 1 else None
    if rank == 0:
        total = B * ws * a.steps
        alg_bytes = (BYTES_IN_CBFQP(K, es) + BYTES_OUT_CBFQP(K, es)) * B
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        res = {
            "metric": "QP solves/sec (batched agents)", "value": total / elapsed, "unit": "solves/s",
            "n_gpus": ws, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.compute, "data": "synthetic",
            "config": {"workload": f"{B}-agent batch DynamicUnicycle2D CBF-QP, {K} obstacles each (BASELINE configs[1])"
                                   if (B, K) == (4096, 8) else f"{B}-agent batch DynamicUnicycle2D CBF-QP, {K} obstacles each",
                       "agents_per_gpu": B, "obstacles": K, "storage": a.io, "arithmetic": a.compute,
                       "launch": "eager" if a.eager else "hipGraph", "sharding": f"agents x{ws}, no collective",
                       "optimal_fraction": n_opt / B},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": ("cbfqp_coop_kernel" if B <= 32768 else "cbfqp_reg_kernel") if K <= 8 else "cbfqp_coop_kernel",
                         "kernel_us": 1e3 * kernel_ms,
                         "algorithmic_bytes_per_solve": BYTES_IN_CBFQP(K, es) + BYTES_OUT_CBFQP(K, es),
                         # HIP events on the launch stream; ms_per_step is the host clock around the same region.  Under
                         # rocprofv3 --kernel-trace the dependent graph nodes are dispatched one completion signal at a time:
                         # its per-dispatch figure is that period when the launch is shorter (profiles/README.md, round 2)
                         "timing": "hip_events_over_timed_region"},
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                res["roofline"]["traffic"] = json.load(open(pmc)).get(f"cbfqp_B{B}_K{K}_{a.io}")
            except Exception:
                pass
        if coll is not None:
            res["collective_leg"] = coll
        if ws == 1 and not a.no_sweep:
            res["sweep"] = sweep(ctl, dev, td, es, K)
            try:
                res["pipelined_two_batches"] = pipelined_leg(ctl, dev, X, ur, ob, K, a.steps)
            except Exception as e:                               # stream-fork capture is an extra, never fatal to the headline
                res["pipelined_two_batches"] = {"error": repr(e)[:200]}
        if ws == 1 and not a.no_mpc:
            res["mpc_cbf"] = mpc_leg(dev, 4096, 8, 10, steps=3, warmup=1,
                                     cpu_seconds=0.0 if a.no_cpu_baseline else 6.0)
            res["od_mpc_cbf"] = od_mpc_leg(dev)
            res["closed_loop"] = closed_loop_leg(dev)
            res["manipulator_cbf_qp"] = manip_leg(dev)
            res["manipulator_closed_loop"] = manip_closed_loop_leg(dev)
            res["quad3d_mpc_cbf"] = linear_mpc_leg(dev, "Quad3D")
            res["single_integrator_mpc_cbf"] = linear_mpc_leg(dev, "SingleIntegrator2D")
            res["quad3d_mpc_cbf_n20"] = linear_mpc_leg(dev, "Quad3D", N=20, steps=2)   # BASELINE config 5's horizon: big layout, four waves per problem
            res["unicycle2d_mpc_cbf"] = unicycle_mpc_leg(dev)
            res["double_integrator_mpc_cbf"] = gn_mpc_leg(dev, "DoubleIntegrator2D")
            res["quad2d_mpc_cbf"] = gn_mpc_leg(dev, "Quad2D")
            res["kinematic_bicycle_mpc_cbf"] = gn_mpc_leg(dev, "KinematicBicycle2D")
            res["kinematic_bicycle_c3bf_mpc_cbf"] = gn_mpc_leg(dev, "KinematicBicycle2D_C3BF")
            try:
                if not NO_LIMIT100:                              # (a counter pass keeps one batch per kernel name)
                    res["c3bf_closed_loop_states_mpc"] = bicycle_loop_leg(dev, "KinematicBicycle2D_C3BF")
                    res["dpcbf_closed_loop_states_mpc"] = bicycle_loop_leg(dev, "KinematicBicycle2D_DPCBF")
            except Exception as e:                               # an extra leg never takes the line down
                res["c3bf_closed_loop_states_mpc"] = {"error": repr(e)[:200]}
            res["vtol_mpc_cbf"] = vtol_mpc_leg(dev)
            res["vtol_ms_mpc_cbf"] = vtol_ms_mpc_leg(dev)
            try:
                if not NO_LIMIT100:
                    res["vtol_reference_scene_closed_loop"] = vtol_ms_closed_loop_leg(dev)
                    res["vtol_reference_scene_fleet_closed_loop"] = vtol_fleet_closed_loop_leg(dev)
            except Exception as e:
                res["vtol_reference_scene_closed_loop"] = {"error": repr(e)[:200]}
            # (od_vtol_mpc_cbf, the condensed optimal-decay VTOL2D kernel -- 1.1 s per 4096, 87 % optimal, superseded by the multiple-shooting
            #  instantiation below -- left the default line in round 6; tests/test_od_vtol_gpu.py still holds it to its oracle)
            res["od_vtol_ms_mpc_cbf"] = od_vtol_ms_mpc_leg(dev)
            res["closed_loop_mpc"] = closed_loop_mpc_leg(dev)
            res["backup_cbf_qp"] = backup_cbf_leg(dev)
            # the two remaining BASELINE configs on the one-GPU line: configs[3] (16384 C3BF agents, one rank: the all-gather is a copy) and
            # configs[4] (the 65536-agent heterogeneous fleet as one warm + one timed step)
            try:
                res["kb_c3bf"] = kb_c3bf_workload(a, dev, ws, rank, backend, collect=True, steps_override=50)
            except Exception as e:
                res["kb_c3bf"] = {"error": repr(e)[:200]}
            try:
                if not NO_LIMIT100:
                    res["hetero_fleet"] = hetero_fleet_workload(a, dev, ws, rank, backend, collect=True)
            except Exception as e:
                res["hetero_fleet"] = {"error": repr(e)[:200]}
        if ws == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(X.double().cpu().numpy(), ur.double().cpu().numpy(),
                                               ob.double().cpu().numpy(), a.cpu_seconds)
        emit(res, ws)
    if ws > 1:
        dist.destroy_process_group()


def hetero_fleet_workload(a, dev, ws, rank, backend, collect=False):
    """BASELINE configs[4], built as SURVEY 8d defines it -- a labelled EXTENSION (no runnable reference): a heterogeneous
    fleet of 65536 agents (or --agents per GPU x ranks when given), half kinematic Unicycle2D and half Quad3D, every
    agent solving an OPTIMAL-DECAY MPC-CBF with horizon 20 against 8 SUPERELLIPSOID obstacles (7-wide rows).  The
    reference's OptimalDecayMPCCBF accepts neither model with decay variables (optimal_decay_mpc_cbf.py:19-20,284-287), so
    the semantics are those of oracle/od_mpc_rd1.py: one decay variable per stage scaling the DT-CBF gain, penalty
    p_sb1 (rho - omega1)^2, r-term R u^2; parity is against that oracle only.  --plain-fleet runs round 1's variant (plain
    MPCCBF of both models on circles).  Agents are independent: contiguous shards per rank, no collective; the two model
    kernels run on two HIP streams of the same GPU."""
    import torch
    import torch.distributed as dist
    import safe_control_amd as sca
    from safe_control_amd import sharding, workloads as W
    n_total = 65536 if a.agents == 4096 else a.agents * ws
    N, K = 20, 8
    lo, hi = sharding.agent_range(n_total // 2, ws, rank)             # the same range of each half
    Bl = hi - lo
    uspec = {"model": "Unicycle2D", "v_max": 1.0, "w_max": 0.5, "radius": 0.25}
    od = not a.plain_fleet
    mi = {"max_iter": a.max_iter} if a.max_iter > 0 else {}
    Xu, gu, _, ou = W.du_cbfqp_batch(n_total // 2, K, seed=0)
    Xq, gq, oq = W.linear_mpc_batch("Quad3D", n_total // 2, K, seed=1)
    if od:
        uni = sca.BatchedOptimalDecayMPCCBF(uspec, io_dtype="f32", horizon=N, extension=True, **mi)
        quad = sca.BatchedOptimalDecayLinearMPCCBF({"model": "Quad3D"}, io_dtype="f32", horizon=N, **mi)
        Xu[:, 3] = 0.0
        ou = W.superellipsoid_obstacles(Xu[:, :2], K, seed=1000)
        oq = W.superellipsoid_obstacles(Xq[:, :2], K, seed=1001)
    else:
        uni = sca.BatchedMPCCBF(uspec, io_dtype="f32", horizon=N, **mi)
        quad = sca.BatchedLinearMPCCBF({"model": "Quad3D"}, io_dtype="f32", horizon=N, **mi)
    t = lambda arr: torch.tensor(arr[lo:hi], dtype=torch.float32, device=dev)
    tXu, tgu, tou = t(Xu), t(gu), t(ou)
    tXq, tgq, toq = t(Xq), t(gq), t(oq)
    upu = torch.zeros((Bl, 2), dtype=torch.float32, device=dev)
    upq = torch.zeros((Bl, 4), dtype=torch.float32, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    res = {}

    def step():
        with torch.cuda.stream(s1):
            res["u"] = uni.solve(tXu, upu, tgu, tou)
        with torch.cuda.stream(s2):
            res["q"] = quad.solve(tXq, upq, tgq, toq)

    torch.cuda.synchronize()
    steps, warm = min(a.steps, 5), min(max(a.warmup, 1), 2)           # a step is 2 x Bl NLPs: seconds, not microseconds
    if collect:                                                        # a leg of the default line: one warm and one timed fleet step
        steps, warm = 1, 1
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    if ws > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if ws > 1:
        dist.barrier()
    elapsed = sharding.max_over_ranks(t1 - t0, device=dev if backend == "nccl" else None)
    if rank == 0:
        su, sq = (res["u"][2], res["q"][2]) if od else (res["u"][1], res["q"][1])
        iu, iq = (res["u"][3], res["q"][3]) if od else (res["u"][2], res["q"][2])
        nbytes = ((16 + 8 + 8 + 7 * K * 4 + 8 + 4 + 4) + (48 + 16 + 12 + 7 * K * 4 + 16 + 4 + 4)) * (n_total // 2)
        if collect:
            return {"workload": f"{n_total}-agent heterogeneous fleet (Unicycle2D + Quad3D), optimal-decay MPC-CBF N=20, 8 superellipsoid obstacles "
                                "(BASELINE configs[4]; extension, oracle/od_mpc_rd1.py)", "extension": True, "agents": n_total,
                    "value": n_total * steps / elapsed, "unit": "solves/s", "steps": steps, "ms_per_step": 1e3 * elapsed / steps,
                    "optimal_fraction": float(((su == 0).double().mean().item() + (sq == 0).double().mean().item()) / 2),
                    "unicycle_optimal_fraction": float((su == 0).double().mean().item()), "quad3d_optimal_fraction": float((sq == 0).double().mean().item()),
                    "max_ipm_iterations": int(max(iu.max().item(), iq.max().item())),
                    "roofline": hetero_roofline(od, n_total, ws, steps, elapsed, nbytes)}
        emit({"metric": "QP solves/sec (batched agents)", "value": n_total * steps / elapsed, "unit": "solves/s",
                          "n_gpus": ws, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * elapsed / steps,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": (f"{n_total}-agent heterogeneous fleet (Unicycle2D + Quad3D), optimal-decay MPC-CBF "
                                                  "N=20, 8 superellipsoid obstacles (BASELINE configs[4]; extension, oracle/od_mpc_rd1.py)")
                                                 if od else
                                                 (f"{n_total}-agent heterogeneous fleet (Unicycle2D + Quad3D), MPC-CBF N=20, "
                                                  "8 circular obstacles (plain MPCCBF of both models)"),
                                     "extension": bool(od),
                                     "agents_total": n_total, "horizon": N, "obstacles": K, "storage": "f32",
                                     "sharding": f"agents x{ws}, no collective; two model kernels on two streams",
                                     "unicycle_optimal_fraction": float((su == 0).double().mean().item()),
                                     "quad3d_optimal_fraction": float((sq == 0).double().mean().item()),
                                     "unicycle_mean_iterations": float(iu.double().mean().item()),
                                     "quad3d_mean_iterations": float(iq.double().mean().item())},
                          "roofline": hetero_roofline(od, n_total, ws, steps, elapsed, nbytes),
                          "cpu_baseline": None}, ws)


def hetero_roofline(od, n_total, ws, steps, elapsed, nbytes):
    """VALU-issue roofline of the fleet step from the committed counter profile of the same 65536-agent launch pair (both
    model kernels run side by side on two streams: their instruction counts add); falls back to the (meaningless) HBM
    figure when the profile does not hold this configuration."""
    if od and n_total == 65536:
        for rnd in ("r06", "r05", "r04", "r03", "r02"):                # the newest round that profiled this launch pair
            path = os.path.join(ROOT, "profiles", f"{rnd}_counters.json")
            try:
                allc = json.load(open(path))
                d = allc.get("hetero_sq", {})
            except Exception:
                allc, d = {}, {}
            # per-dispatch averages x dispatches of one fleet step (a budget solve of either kernel = classify + cap 100 + rest)
            insts = [c["SQ_INSTS_VALU"] * LAUNCHES_PER_BUDGET_SOLVE for k, c in d.items()
                     if ("mpclin_kernel" in k or "odmpccbf_uni_kernel" in k) and "SQ_INSTS_VALU" in c]
            if len(insts) == 2:
                ach = sum(insts) * steps / elapsed / 1e9
                return {"bound": "valu_issue", "achieved": ach, "peak": VALU_PEAK_GIPS * ws, "unit": "G wave-instr/s", "frac": ach / (VALU_PEAK_GIPS * ws),
                        "traffic": None, "kernel": "odmpccbf_uni_kernel<20> + mpclin_kernel<12,4,0,0,big,od>", "valu_instructions_per_step": sum(insts),
                        "source": f"profiles/{rnd}_counters.json:hetero_sq", "stale": allc.get("_meta", {}).get("csrc_sha16") != csrc_sha16(),
                        "note": "latency-bound interior-point solves (two Quad3D problems per CU: 77 KB of LDS and four waves each; two Unicycle2D problems per CU: 65 KB, one wave each)"}
    return {"bound": "hbm", "achieved": nbytes * steps / elapsed / 1e9, "peak": HBM_PEAK_GBS * ws, "unit": "GB/s",
            "frac": nbytes * steps / elapsed / 1e9 / (HBM_PEAK_GBS * ws), "traffic": None,
            "kernel": ("odmpccbf_uni_kernel<20> + mpclin_kernel<12,4,0,0,big,od>" if od else "mpccbf_uni_kernel<20> + mpclin_kernel<12,4,0,0,big>"),
            "note": "interior-point solves: VALU / latency bound, HBM bytes are negligible (no VALU counter profile committed for this configuration)"}


def kb_c3bf_workload(a, dev, ws, rank, backend, collect=False, steps_override=None):
    """BASELINE configs[3]: 16384 KinematicBicycle2D C3BF agents in total, sharded over the ranks (strong scaling);
    a step = all-gather of the agent states (RCCL over xGMI; a no-op on one rank) -> K = 16 nearest other agents as
    moving circular obstacles (neighbour kernel) -> C3BF CBF-QP for the local shard."""
    import torch
    import torch.distributed as dist
    import safe_control_amd as sca
    from safe_control_amd import sharding, workloads as W
    n_agents, K = 16384, 16
    spec = {"model": "KinematicBicycle2D_C3BF", "a_max": 5.0, "radius": 0.3}
    ctl = sca.BatchedCBFQP(dict(spec), dt=0.05, io_dtype="f32", compute_dtype="f64")
    Xn, goal, un, on = W.kb_c3bf_batch(n_agents, K, seed=0, spec=spec)
    side = 3.0 * n_agents ** 0.5 / 14.0                              # spread the fleet: ~3 m mean spacing (the 14 m field of
    Xn[:, 0:2] *= side; goal = goal * side                           # SURVEY 8d would put 84 agents on every square metre)
    from safe_control_amd.robots.spec import complete_robot_spec
    un = W.nominal_input_kb(Xn, goal, complete_robot_spec(dict(spec)))
    lo, hi = sharding.agent_range(n_agents, ws, rank)
    X = torch.tensor(Xn[lo:hi], dtype=torch.float32, device=dev)
    ur = torch.tensor(un[lo:hi], dtype=torch.float32, device=dev)
    Bl = hi - lo
    out = (torch.empty((Bl, 2), dtype=torch.float32, device=dev), torch.empty((Bl,), dtype=torch.int32, device=dev),
           torch.empty((Bl, K), dtype=torch.float32, device=dev))

    ex = sharding.NeighborExchange(n_agents, K, 0.3, nx=4, dtype=torch.float32, device=dev)   # persistent buffers

    def step():
        ctl.solve(X, ur, ex.step(X), out=out)

    steps = steps_override or a.steps
    for _ in range(max(a.warmup, 1)):
        step()
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if ws > 1:
        dist.barrier()
    elapsed = sharding.max_over_ranks(t1 - t0, device=dev if backend == "nccl" else None)
    if collect:
        nb = (16 + 8 + 7 * K * 4 + 8 + 4 + K * 4) * n_agents
        return {"workload": "16384-agent KinematicBicycle2D C3BF, 16 nearest other agents as moving obstacles (BASELINE configs[3]): "
                            "all-gather of the states (RCCL) -> neighbour kernel -> CBF-QP per step",
                "value": n_agents * steps / elapsed, "unit": "solves/s", "steps": steps, "ms_per_step": 1e3 * elapsed / steps,
                "scaling": "strong", "agents": n_agents, "agents_total": n_agents, "all_gather_bytes_per_step": n_agents * 16,
                "equal_shards": bool(ex.equal), "achieved_GBs": nb * steps / elapsed / 1e9,
                "optimal_fraction": float((out[1] == 0).double().mean().item()),
                "roofline": {"bound": "hbm", "achieved": nb * steps / elapsed / 1e9, "peak": HBM_PEAK_GBS * ws, "unit": "GB/s",
                             "frac": nb * steps / elapsed / 1e9 / (HBM_PEAK_GBS * ws), "traffic": None,
                             "kernel": "nb_bbox / nb_count / nb_scan / nb_scatter / nb_select (uniform-grid cell list) + cbfqp kernel: six launches per step",
                             "algorithmic_bytes_per_solve": nb // n_agents}}
    if rank == 0:
        nbytes = (16 + 8 + 7 * K * 4 + 8 + 4 + K * 4) * n_agents
        emit({"metric": "QP solves/sec (batched agents)", "value": n_agents * a.steps / elapsed, "unit": "solves/s",
                          "n_gpus": ws, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": "16384-agent KinematicBicycle2D C3BF, 16 nearest other agents as moving "
                                                 "obstacles (BASELINE configs[3])", "agents_total": n_agents, "obstacles": K,
                                     "storage": "f32", "sharding": f"agents x{ws}, all-gather of the states per step",
                                     "optimal_fraction": float((out[1] == 0).double().mean().item())},
                          "roofline": {"bound": "hbm", "achieved": nbytes * a.steps / elapsed / 1e9, "peak": HBM_PEAK_GBS * ws,
                                       "unit": "GB/s", "frac": nbytes * a.steps / elapsed / 1e9 / (HBM_PEAK_GBS * ws),
                                       "traffic": None, "kernel": "neighbor_kernel + cbfqp_coop_kernel",
                                       "algorithmic_bytes_per_solve": nbytes // n_agents},
                          "cpu_baseline": None}, ws)


def sweep(ctl, dev, td, es, K):
    """Same kernel at batch sizes where HBM is the binding limit (inputs drawn on-device)."""
    import math
    import torch
    import safe_control_amd as sca
    from safe_control_amd import _lib
    ctl32 = None
    if ctl.compute_dtype != _lib.DTYPE_F32 and ctl.io_dtype == _lib.DTYPE_F32:
        ctl32 = sca.BatchedCBFQP(dict(ctl.robot_spec), dt=ctl.dt, io_dtype="f32", compute_dtype="f32")
    res = []
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    for B in (65536, 1 << 20, 1 << 24):
        X = torch.rand((B, 4), generator=g, device=dev, dtype=torch.float32)
        X[:, 0:2] *= 14.0
        X[:, 2] = (X[:, 2] * 2 - 1) * math.pi
        goal = torch.rand((B, 2), generator=g, device=dev) * 14.0
        r = torch.rand((B, K), generator=g, device=dev) * 0.8 + 0.2
        rho = torch.rand((B, K), generator=g, device=dev) * (4.0 - (r + 0.3)) + (r + 0.3)
        phi = (torch.rand((B, K), generator=g, device=dev) * 2 - 1) * math.pi
        obs = torch.zeros((B, K, 7), device=dev)
        obs[:, :, 0] = X[:, None, 0] + rho * torch.cos(phi)
        obs[:, :, 1] = X[:, None, 1] + rho * torch.sin(phi)
        obs[:, :, 2] = r
        err = torch.remainder(torch.atan2(goal[:, 1] - X[:, 1], goal[:, 0] - X[:, 0]) - X[:, 2] + math.pi, 2 * math.pi) - math.pi
        d = (torch.hypot(X[:, 0] - goal[:, 0], X[:, 1] - goal[:, 1]) - 0.05).clamp_min(0)
        v = torch.where(err.abs() > math.pi / 2, torch.zeros_like(d), (d * torch.cos(err)).clamp_max(1.0))
        ur = torch.stack([v - X[:, 3], 2.0 * err], dim=1)
        X, ur, obs = X.to(td).contiguous(), ur.to(td).contiguous(), obs.to(td).contiguous()
        out = (torch.empty((B, 2), dtype=td, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
               torch.empty((B, K), dtype=td, device=dev))
        nbytes = (BYTES_IN_CBFQP(K, es) + BYTES_OUT_CBFQP(K, es)) * B
        row = {"agents": B}
        # the arithmetic of the headline line first, then (same storage) f32 arithmetic where HBM is the tighter bound
        for label, c in (("", ctl), ("_f32_arithmetic", ctl32)):
            if c is None:
                continue
            for _ in range(3):
                c.solve(X, ur, obs, out=out)
            torch.cuda.synchronize()
            n = 20 if B <= (1 << 20) else 10
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                c.solve(X, ur, obs, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            gbs = nbytes / (ms * 1e-3) / 1e9
            row.update({"kernel_us" + label: 1e3 * ms, "solves_per_s" + label: B / (ms * 1e-3),
                        "achieved_GBs" + label: gbs, "frac_hbm_peak" + label: gbs / HBM_PEAK_GBS,
                        "frac_hbm_achievable" + label: gbs / HBM_ACHIEVABLE_GBS})
        res.append(row)
        del X, ur, obs, out
    return res


if __name__ == "__main__":
    main()
