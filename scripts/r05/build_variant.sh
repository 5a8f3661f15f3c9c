#!/bin/bash
# usage: scripts/r05/build_variant.sh <name> [-DFOO=1 ...]  -> scripts/r05/libsparrow_hip_<name>.so (git-ignored; travels to the GPU box)
cd "$(dirname "$0")/../.."
n=$1; shift
# (one unit: -DSPRK_SINGLE_TU folds the kernel-family units back into sparrow_hip.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -DSPRK_SINGLE_TU "$@" -I include -I sparrowrecsys_amd/csrc sparrowrecsys_amd/csrc/sparrow_hip.hip -o scripts/r05/libsparrow_hip_$n.so && echo "built $n"
