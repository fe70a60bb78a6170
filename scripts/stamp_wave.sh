#!/bin/bash
# Diagnostic build of the library with s_memtime stamps in k_apply_wave (MIMSEM_STAMPS), built ON THE GPU BOX in its ephemeral copy
# of the repo, then one Umat apply hot (103 680 units) and cold (8 spheres): where a wavefront's lifetime goes.
cd $GRAFT_REPO_ROOT/mimsem_amd/csrc && make clean > /dev/null && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMIMSEM_STAMPS" > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
cd $GRAFT_REPO_ROOT && REPS=3 MIMSEM_WAVE_STAMPS=1 python3 scripts/prof_umat.py 2>&1 | grep -v "^$" | tail -60
