#!/bin/bash
# builds experimental variants of the library next to the real one (timing experiments only):
#   tools/build_ablate.sh NAME:"-DFOO -DBAR=2" ...
set -e
cd "$(dirname "$0")/../vfm-registration_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
for spec in "$@"; do
  v="${spec%%:*}"; defs="${spec#*:}"
  # the ablation switches (-DVFM_ABLATE_FOLD / _DMA / _LDS) live in the fp16 coarse kernels
  hipcc $F $defs -c csrc/match_coarse_f16.hip -o build/match_coarse_f16_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o vfmreg/lib/libvfmreg_hip_$v.so build/error.cpp.o build/config.cpp.o build/match_api.hip.o build/match_prep.hip.o \
        build/match_coarse_f16_$v.o build/match_coarse_i8.hip.o build/match_finish.hip.o build/match_l2.hip.o build/ransac.hip.o \
        build/project.hip.o build/vit.hip.o build/vit_mlp.hip.o build/icp.hip.o build/voxel.hip.o
done
