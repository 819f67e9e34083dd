#!/bin/bash
# Builds the experiment library of tools/experiments/level0_lds.sh in the build container (hipcc cross-compiles gfx950):
# csrc/field_eval.hip + level0_lds.patch with -DQF_L0_LDS, misc.cpp with the experiment ABI offset, the other objects of
# the product build.  Output: tools/experiments/_build/libqf_l0lds.so (git-ignored; travels to the GPU box with gpurun).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/tools/experiments/_build
mkdir -p $B
python3 -m quadraturefields_amd.build > /dev/null          # the product objects (csrc/_obj)
cp $R/quadraturefields_amd/csrc/field_eval.hip $B/field_eval.hip
(cd $B && patch -p3 < $R/tools/experiments/level0_lds.patch)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$R/quadraturefields_amd/csrc"
/opt/rocm/bin/hipcc $F -DQF_L0_LDS -c $B/field_eval.hip -o $B/field_eval_l0.o
/opt/rocm/bin/hipcc $F -DQF_ABI_VERSION_OFFSET=1000 -x hip -c $R/quadraturefields_amd/csrc/misc.cpp -o $B/misc_exp.o
OBJS=$(ls $R/quadraturefields_amd/csrc/_obj/*.o | grep -v "field_eval.o\|misc.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $B/libqf_l0lds.so $B/field_eval_l0.o $B/misc_exp.o $OBJS
echo built $B/libqf_l0lds.so
