#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/collect_profiles.sh r01
# Writes rocprofv3 summaries under gpurun_out/profiles_<round>/ ; tools/parse_profiles.py turns them
# into the committed files under profiles/.
set -u
ROUND=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-sweep --no-mpc"
# 1. kernel trace + stats of the bench command (timed region = 200 launches of the B=4096 workload)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2>/dev/null
# 2. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o bench_fetch -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o bench_write -- $BENCH > /dev/null 2>&1
# 3. same counters on large batches (calibration of the counter against known algorithmic bytes, and the
#    regime where HBM is the binding limit): 2^20 and 2^23 agents, f32 storage, f32 and f64 arithmetic
for CFG in "1048576 8 f32 f32" "1048576 8 f32 f64" "8388608 8 f32 f32" "4096 8 f32 f64"; do
  TAG=$(echo $CFG | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o big_$TAG -- python3 $R/tools/prof_cbfqp.py $CFG 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o bigf_$TAG -- python3 $R/tools/prof_cbfqp.py $CFG 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o bigw_$TAG -- python3 $R/tools/prof_cbfqp.py $CFG 6 > /dev/null 2>&1
done
# 4. SQ counters of the dominant kernel at 2^20 agents (f32/f32 and f32/f64)
for CFG in "1048576 8 f32 f32" "1048576 8 f32 f64"; do
  TAG=$(echo $CFG | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o sqa_$TAG -- python3 $R/tools/prof_cbfqp.py $CFG 4 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sqb_$TAG -- python3 $R/tools/prof_cbfqp.py $CFG 4 > /dev/null 2>&1
done
# 4b. the full default bench line (sweep, closed-loop rollout and MPC legs) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench_full -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_full_under_rocprof.json 2>/dev/null
# 4c. instruction counts of every interior-point leg of the default line (VALU rooflines of bench.py: with_roofline)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o bench_full_sq -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-limit100 > /dev/null 2>&1
# 4d. lane utilisation and the f64 instruction mix of the same legs (separate passes: the SQ counters of one pass are limited)
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT -o bench_full_lane -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-limit100 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT -o bench_full_flop -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-limit100 > /dev/null 2>&1
# 5. MPC-CBF kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o mpc -- python3 $R/bench.py --workload mpc_cbf --steps 5 --warmup 1 > $OUT/mpc_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o mpc_sq -- python3 $R/bench.py --workload mpc_cbf --steps 2 --warmup 1 --no-limit100 --no-cpu-baseline > /dev/null 2>&1
# 6. linear-model MPC-CBF kernel (Quad3D, n = 40): instruction mix and LDS conflicts
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o mpclin -- python3 $R/tools/prof_mpclin.py Quad3D 4096 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o mpclin_sq -- python3 $R/tools/prof_mpclin.py Quad3D 4096 2 > /dev/null 2>&1
# 7. the same kernel's big layout (Quad3D at N = 20, n = 80: four waves per problem): kernel trace, instruction mix, I-cache
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o mpclin_big -- python3 $R/tools/prof_mpclin.py Quad3D 4096 2 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o mpclin_big_sq -- python3 $R/tools/prof_mpclin.py Quad3D 1024 2 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT -o mpclin_big_ic -- python3 $R/tools/prof_mpclin.py Quad3D 1024 2 20 > /dev/null 2>&1
# 8. heterogeneous fleet, BASELINE configs[4] as stated (optimal decay + superellipsoids, extension) and round 1's plain variant
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o hetero -- python3 $R/bench.py --workload hetero_fleet --steps 1 --warmup 1 > $OUT/hetero_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o hetero_plain -- python3 $R/bench.py --workload hetero_fleet --plain-fleet --steps 1 --warmup 1 > $OUT/hetero_plain_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o hetero_sq -- python3 $R/bench.py --workload hetero_fleet --steps 1 --warmup 1 > /dev/null 2>&1
# 9. Backup-CBF QP kernel: instruction mix
python3 $R/tools/prof_backup.py 4096 3 --prepare > /dev/null 2>&1     # the fleet (closed-loop rollouts) outside the profiled process
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o backup_sq -- python3 $R/tools/prof_backup.py 4096 3 > /dev/null 2>&1
# 11. VTOL2D MPC-CBF kernel (one NLP per wavefront): kernel trace, instruction mix, LDS conflicts

rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o vtol -- python3 $R/tools/time_mpcvtol.py 4096 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o vtol_sq -- python3 $R/tools/time_mpcvtol.py 4096 > /dev/null 2>&1
# 11b. what the VTOL2D kernel's spilled registers cost: scratch (FLAT / VMEM) instruction counts, waits, HBM traffic per launch
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT -o vtol_mem -- python3 $R/tools/time_mpcvtol.py 4096 f32 limit100 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT -o vtol_mem2 -- python3 $R/tools/time_mpcvtol.py 4096 f32 limit100 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o vtol_fetch -- python3 $R/tools/time_mpcvtol.py 4096 f32 limit100 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o vtol_write -- python3 $R/tools/time_mpcvtol.py 4096 f32 limit100 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $OUT -o vtol_tcp -- python3 $R/tools/time_mpcvtol.py 4096 f32 limit100 > /dev/null 2>&1
# 12. the multiple-shooting VTOL2D kernel of round 5 (csrc/mpc_vtol_ms.hip): kernel trace, instruction mix, scratch traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o vtolms -- python3 $R/tools/time_mpcvtol.py 4096 f32 ms > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o vtolms_sq -- python3 $R/tools/time_mpcvtol.py 4096 f32 ms > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT -o vtolms_mem -- python3 $R/tools/time_mpcvtol.py 4096 f32 ms > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT -o vtolms_mem2 -- python3 $R/tools/time_mpcvtol.py 4096 f32 ms > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o vtolms_fetch -- python3 $R/tools/time_mpcvtol.py 4096 f32 ms > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o vtolms_write -- python3 $R/tools/time_mpcvtol.py 4096 f32 ms > /dev/null 2>&1
# 13. BASELINE configs[3] on one rank: the neighbour search (uniform-grid cell list: nb_bbox / nb_count / nb_scan / nb_scatter / nb_select) + C3BF CBF-QP per step;
#     and the search next to the plain scan it is held to (neighbor_kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kbc3bf -- python3 $R/bench.py --workload kb_c3bf --steps 50 --warmup 5 > $OUT/kbc3bf_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o kbc3bf_sq -- python3 $R/bench.py --workload kb_c3bf --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o kbc3bf_fetch -- python3 $R/bench.py --workload kb_c3bf --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o kbc3bf_write -- python3 $R/bench.py --workload kb_c3bf --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o neighbors -- python3 $R/tools/time_neighbors.py > /dev/null 2>&1
# 14. kernel 13 (csrc/mpc_du_ms.hip: configs[2] as do-mpc poses it) next to the condensed kernel on the same batch: kernel trace, instruction mix, memory side
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dums -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o dums_sq -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT -o dums_mem -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT -o dums_mem2 -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o dums_fetch -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o dums_write -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 > /dev/null 2>&1
# ... and its DoubleIntegrator2D instantiation
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dumsdi -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 3 di > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o dumsdi_sq -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 di > /dev/null 2>&1
# ... Unicycle2D (general stage layout, one-step rows)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dumsuni -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 3 uni > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o dumsuni_sq -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 uni > /dev/null 2>&1
# ... SingleIntegrator2D
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o dumssi_sq -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 si > /dev/null 2>&1
# ... and KinematicBicycle2D (general stage layout; with the reference solver's budget: a few solves cycle to it)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dumskb -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 2 kb > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o dumskb_sq -- python3 $R/tools/time_mpcdu_ms.py 4096 f32 1 kb > /dev/null 2>&1
# 10. sustained VALU issue peak of the part (the denominator of the valu_issue rooflines)
[ -x $R/exp_libs/valu_peak ] || { mkdir -p $R/exp_libs; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/micro/valu_peak.hip -o $R/exp_libs/valu_peak > /dev/null 2>&1; }
[ -x $R/exp_libs/valu_peak ] && $R/exp_libs/valu_peak > $OUT/valu_peak.txt 2>&1
ls $OUT | head -80
