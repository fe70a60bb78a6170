#!/bin/bash
# Build another variant of the library next to the default one: scripts/build_variant.sh NAME "-DFLAG ..." -> build_ab/libmimsem_hip_NAME.so
# (select it with MIMSEM_LIB=$PWD/build_ab/libmimsem_hip_NAME.so; build_ab/ is git-ignored and travels to the GPU box)
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build_ab/obj_$name; mkdir -p $d
for f in api elem_kernels column_kernels krylov_kernels halo ksp; do
  if [ ! -f $d/$f.o ] || [ -n "$(find $root/mimsem_amd/csrc $root/include -newer $d/$f.o \( -name '*.hip' -o -name '*.inc' -o -name '*.hpp' -o -name '*.h' \) | head -1)" ]; then      # (any source newer than the object: rebuild it)
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable $flags -c $root/mimsem_amd/csrc/$f.hip -o $d/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/build_ab/libmimsem_hip_$name.so $d/api.o $d/elem_kernels.o $d/column_kernels.o $d/krylov_kernels.o $d/halo.o $d/ksp.o -ldl
echo built $root/build_ab/libmimsem_hip_$name.so
