"""Rewrites the "Round 4 files" section of profiles/README.md and profiles/r04_uniform_batches.json from the parsed r04 files
(tools/parse_profiles.py r04 first) and the uniform-batch counter passes (gpurun_out/profiles_r04u, tools/collect_uniform.sh).
The unprofiled uniform-batch launch times come from gpurun_out/bench_legs.json (a default bench.py run) when it holds them."""
import csv, json, collections, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(f'{ROOT}/profiles/r04_kernel_stats.csv')))
def fmt(ns):
    ns = float(ns)
    return f"{ns/1e6:.2f} ms" if ns >= 1e6 else f"{ns/1e3:.2f} µs"
tab = '\n'.join(f"| `{r['run']}` | `{r['Name'].replace('void sc::','')}` | {r['Calls']} | {fmt(r['AverageNs'])} | {fmt(r['MinNs'])} |"
                for r in rows if r['run'] in ('bench_full', 'bench', 'mpc', 'mpclin', 'mpclin_big', 'vtol', 'hetero', 'hetero_plain'))
c = json.load(open(f'{ROOT}/profiles/r04_counters.json'))
sq = c['bench_full_sq']
lines = [f"  | `{k.replace('void sc::','')}` | {v['SQ_INSTS_VALU']/1e6:.1f} M | {v.get('SQ_INSTS_SALU',0)/max(v['SQ_INSTS_VALU'],1):.2f} | {v.get('SQ_ACTIVE_INST_VALU',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f} |"
         for k, v in sorted(sq.items()) if 'SQ_INSTS_VALU' in v and any(x in k for x in ('mpc', 'odmpc', 'backup'))]
def one(run, ctr):
    for k, v in c.get(run, {}).items():
        if ctr in v:
            return v[ctr]
def conf(run):
    for k, v in c.get(run, {}).items():
        if 'SQ_LDS_BANK_CONFLICT' in v and v.get('SQ_ACTIVE_INST_LDS'):
            return v['SQ_LDS_BANK_CONFLICT'] / v['SQ_ACTIVE_INST_LDS'], v['SQ_ACTIVE_INST_LDS'] / v['SQ_WAVE_CYCLES']
cm, cl, cv = conf('mpc_sq'), conf('mpclin_sq'), conf('vtol_sq')
vm = lambda ctr: one('vtol_mem', ctr)
fetch, write = one('vtol_fetch', 'FETCH_SIZE') * 1024 / 1e9, one('vtol_write', 'WRITE_SIZE') * 1024 / 1e9
waitfrac = one('vtol_mem2', 'SQ_WAIT_ANY') / one('vtol_mem', 'SQ_WAVE_CYCLES')
tot_inst = one('vtol_sq', 'SQ_INSTS_VALU') + one('vtol_sq', 'SQ_INSTS_SALU') + one('vtol_sq', 'SQ_INSTS_LDS') + vm('SQ_INSTS_FLAT')
bench_uniform = {'du': (1.1138, 18), 'kb': (6.842, 18), 'c3bf': (9.936, 17), 'quad3d': (1.3653, 6), 'vtol': (28.900, 31)}
try:
    b = json.load(open(f'{ROOT}/gpurun_out/bench_legs.json'))
    m = {'du': 'mpc_cbf', 'kb': 'kinematic_bicycle_mpc_cbf', 'c3bf': 'kinematic_bicycle_c3bf_mpc_cbf', 'quad3d': 'quad3d_mpc_cbf', 'vtol': 'vtol_mpc_cbf'}
    for f, k in m.items():
        ub = b[k]['uniform_batch']; bench_uniform[f] = (ub['kernel_ms'], ub['iterations'])
except Exception as e:
    print('bench_legs.json not usable, keeping the recorded uniform-batch times:', e)
U = {}
for f in ('du', 'kb', 'c3bf', 'quad3d', 'vtol'):
    rr = list(csv.DictReader(open(f'{ROOT}/gpurun_out/profiles_r04u/uni_{f}_counter_collection.csv')))
    byd = collections.OrderedDict()
    for r in rr:
        if 'mpc' not in r['Kernel_Name']:
            continue
        byd.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
        byd[int(r['Dispatch_Id'])]['k'] = r['Kernel_Name'].split('(')[0]
    d = list(byd.values())
    note = open(f'{ROOT}/gpurun_out/profiles_r04u/uni_{f}.txt').read().strip()
    it_prof = int(note.split(' iterations each')[0].split()[-1])
    ms, itb = bench_uniform[f]
    insts = d[-1]['SQ_INSTS_VALU'] * itb / it_prof
    U[f] = {'kernel': d[-1]['k'].replace('void sc::', ''), 'real_batch_SQ_INSTS_VALU': d[0]['SQ_INSTS_VALU'], 'uniform_batch_SQ_INSTS_VALU': d[-1]['SQ_INSTS_VALU'],
            'uniform_batch_iterations': it_prof, 'uniform_SQ_ACTIVE_INST_VALU_over_WAVE_CYCLES': d[-1]['SQ_ACTIVE_INST_VALU'] / d[-1]['SQ_WAVE_CYCLES'],
            'bench_uniform_kernel_ms': ms, 'bench_uniform_iterations': itb, 'valu_issue_frac_direct': insts / (ms * 1e-3) / 1e9 / 519.0, 'tool_output': note}
U['_meta'] = {'how': 'tools/collect_uniform.sh (rocprofv3 --pmc SQ_INSTS_VALU ... -- tools/prof_uniform.py FAMILY): instructions of ONE launch of the uniform batch, counted; '
                     'launch time from bench.py without the profiler (uniform_batch.kernel_ms); peak 519 G wave-instr/s', 'csrc_sha16': c['_meta']['csrc_sha16']}
json.dump(U, open(f'{ROOT}/profiles/r04_uniform_batches.json', 'w'), indent=1)
a = lambda f: U[f]['uniform_SQ_ACTIVE_INST_VALU_over_WAVE_CYCLES']
d_ = lambda f: U[f]['valu_issue_frac_direct']
nl = chr(10)
sec = f'''## Round 4 files (`r04_*`; `tools/collect_profiles.sh r04`, collected after the last kernel change of the round: csrc hash `{c['_meta']['csrc_sha16']}`)

Same commands as round 3, plus the VTOL2D memory passes (`vtol_mem`, `vtol_mem2`, `vtol_fetch`, `vtol_write`, `vtol_tcp`) and the
uniform-batch passes (`tools/collect_uniform.sh` -> `r04_uniform_batches.json`); this section is written by
`tools/profiles_round4_section.py`.  **What is new in reading them:** the interior-point
classes run the reference solver's budget as continuation launches, so ONE solve of a batch is up to three dispatches of the same
kernel -- a classify-only pre-pass (~4 µs: the `Min` column), the launch to the cap of 100 and the launch that finishes the stragglers
-- and the per-kernel averages below are per DISPATCH over that mix (plus, in `bench_full`, the one-launch limit-100 comparison and the
uniform batch each leg times, and for `mpcgn_kernel<2 | 3, ...>` the ~170 launches of the closed-loop fleets that prepare the
closed-loop-state legs).  The counter passes behind the VALU-issue rooflines (`bench_full_sq`, `_lane`, `_flop`) run
`bench.py --no-limit100` without those legs: one kind of solve per kernel name, and `bench.py` multiplies the per-dispatch
average by three.

| run | kernel | dispatches | average | minimum |
|---|---|---|---|---|
{tab}

* `bench`: the headline launch, unchanged: 2.56 µs minimum, ~4 µs instrumented dispatch period (3.0 µs by HIP events without the
  profiler).
* `hetero` (BASELINE configs[4], extension) with the budget behind the optimal-decay kernels (one launch each): the fleet step is as
  long as the fleet's longest solve (0.43 / 0.46 s at the 100-iteration limit of rounds 2 - 3: `bench.py --workload hetero_fleet
  --max-iter 100`).
* `bench_full_sq`: VALU wave-instructions per dispatch, SALU / VALU and the VALU-active share of the wave cycles of the interior-point
  kernels (`bench.py` turns the first column into the `valu_issue` fractions of the line, against 519 G wave-instr/s):

  | kernel | VALU instr / dispatch | SALU / VALU | VALU active / wave cycles |
  |---|---|---|---|
{nl.join(lines)}

* LDS bank conflicts (`mpc_sq`, `mpclin_sq`, `vtol_sq`: `SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS`) are {cm[0]:.2f} for `mpccbf_kernel<10, 8>`, {cl[0]:.2f} for
  `mpclin_kernel<12, 4, 10, 8>` and {cv[0]:.2f} for the VTOL2D kernel -- of an LDS pipe that is active for **4 - 5 % of the wave cycles**
  (`SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES`: {cm[1]:.3f} / {cl[1]:.3f} / {cv[1]:.3f}).  Removing every conflict would buy 2 - 4 %; the VALU is active for
  a third of the wave cycles and the rest are dependency stalls of one or two waves per SIMD (the occupancy is set by LDS:
  20 KB per config-3 problem = 8 per CU, 49 - 54 KB per `mpcgn` problem = 3 per CU, where a wave runs an iteration in the same 143 k
  cycles alone and in a full machine: `tools/exp_mpcgn_phases.py`, 256 and 4096 problems).
* **Work level, counted directly** (`r04_uniform_batches.json`; `tools/collect_uniform.sh`: one launch of the batch filled with copies
  of the median-iteration problem under `--pmc SQ_INSTS_VALU ...`, launch time from the unprofiled `bench.py` run): VALU-issue fraction
  **config 3 {d_('du'):.2f}, KinematicBicycle2D {d_('kb'):.2f}, C3BF {d_('c3bf'):.2f}, Quad3D {d_('quad3d'):.2f}, VTOL2D {d_('vtol'):.2f}** (`bench.py` derives its `work_frac` from
  the mixed batch's instructions per iteration: within 20 %).  Per WAVE the VALU is active for {a('du'):.2f} (config 3), {a('kb'):.2f} (KB), {a('c3bf'):.2f} (C3BF),
  {a('quad3d'):.2f} (Quad3D), {a('vtol'):.2f} (VTOL2D) of its cycles; the chip-level fraction is that times the waves per SIMD the LDS footprint allows --
  2 for config 3 (20 KB per problem), 0.75 for `mpcgn` (49 - 54 KB), 0.5 for Quad3D N = 10 (75 KB), 1 for VTOL2D (39 KB, 512 VGPRs).
  Occupancy, i.e. LDS per problem, is what separates config 3 from the rest at the work level.
* **What the VTOL2D kernel's spilled registers cost** (review item 3 asked for the measurement; `mpcvtol_wave_kernel<float, 8, false>`,
  4096 problems, ONE launch at the 100-iteration limit, 61 ms; 512 VGPRs + 1020 spilled, 2784 B of scratch per lane):
  * {one('vtol_sq','SQ_INSTS_VALU')/1e9:.2f} G VALU, {one('vtol_sq','SQ_INSTS_SALU')/1e9:.2f} G SALU, {one('vtol_sq','SQ_INSTS_LDS')/1e9:.2f} G LDS and **{vm('SQ_INSTS_FLAT')/1e9:.3f} G scratch (FLAT) wave-instructions** per launch ({vm('SQ_INSTS_VMEM_RD')/1e6:.1f} M loads, {vm('SQ_INSTS_VMEM_WR')/1e6:.1f} M stores): one
    instruction in {tot_inst/vm('SQ_INSTS_FLAT'):.0f} (of VALU + SALU + LDS + FLAT) is a spill access -- about 880 per interior-point iteration and wave.
  * They do not stay in the caches: **FETCH_SIZE {fetch:.1f} GB and WRITE_SIZE {write:.1f} GB per launch** (x 2 on the fetch side with the gfx950
    correction of the guide: ~{2*fetch+write:.0f} GB) against 1.2 MB of algorithmic input and output -- 1024 resident waves x 64 lanes x 2.7 KB = 177 MB of
    scratch is more than the 32 MB of L2.  That is 0.6 - 0.8 TB/s, under a tenth of the HBM peak: the kernel is not bandwidth-bound, it
    waits -- `SQ_WAIT_ANY` is {100*waitfrac:.0f} % of the wave cycles (`vtol_mem2`), and with one wave per SIMD there is nothing to switch to.
  * `TCP_TOTAL_CACHE_ACCESSES` {one('vtol_tcp','TCP_TOTAL_CACHE_ACCESSES_sum')/1e9:.1f} G per launch, {one('vtol_tcp','TCP_TCC_READ_REQ_sum')/1e9:.2f} G read and {one('vtol_tcp','TCP_TCC_WRITE_REQ_sum')/1e9:.2f} G write requests to L2 (`vtol_tcp`).
  A spill-free layout needs the row state of a stage split over two lanes (DESIGN.md (f)); not built this round.
'''
p = f'{ROOT}/profiles/README.md'
s = open(p).read()
i = s.index('## Round 4 files')
open(p, 'w').write(s[:i].rstrip('\n') + '\n\n' + sec)
print({k: round(v['valu_issue_frac_direct'], 3) for k, v in U.items() if k != '_meta'})
