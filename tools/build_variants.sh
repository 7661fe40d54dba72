#!/bin/bash
# Builds variants of ONE translation unit of the HIP library into exp_libs/lib_<name>.so (the other objects come from build/csrc):
#   tools/build_variants.sh mpc_gn  L0="-DSC_CONT_LEVEL=0" O2="-O2" ...
# Run a tool against one with SC_EXP_LIB=exp_libs/lib_<name>.so.
set -e
cd "$(dirname "$0")/.."
tu=$1; shift
mkdir -p exp_libs build/var
others=$(ls build/csrc/*.o | grep -v "/$tu.o")
pids=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-pass-failed -w $flags \
        -c safe_control_amd/csrc/$tu.hip -o build/var/${tu}_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp_libs/lib_$name.so build/var/${tu}_$name.o $others && echo "built $name"
  ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
