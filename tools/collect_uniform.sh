cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_r04u; mkdir -p $OUT
for f in du kb c3bf quad3d vtol; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o uni_$f -- python3 $R/tools/prof_uniform.py $f 3 > $OUT/uni_$f.txt 2>/dev/null
  cat $OUT/uni_$f.txt
done
ls $OUT | head -30
