#!/bin/bash
# builds experimental variants of the library next to the real one (timing experiments only):
#   tools/build_ablate.sh NAME:"-DFOO -DBAR=2" ...
set -e
cd "$(dirname "$0")/../vfm-registration_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
for spec in "$@"; do
  v="${spec%%:*}"; defs="${spec#*:}"
  hipcc $F $defs -c csrc/match.hip -o build/match_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o vfmreg/lib/libvfmreg_hip_$v.so build/error.cpp.o build/match_$v.o build/ransac.hip.o build/project.hip.o build/vit.hip.o build/icp.hip.o build/voxel.hip.o
done
