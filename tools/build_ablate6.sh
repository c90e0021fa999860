#!/bin/bash
# builds timing-experiment variants of the fp6 coarse kernel next to the real library (results of such builds are garbage):
#   tools/build_ablate6.sh NAME:"-DVFM_ABL_NOFOLD -DVFM_ABL_NOLDS" ...   ->  vfmreg/lib/libvfmreg_hip_NAME.so
# switches (csrc/match_coarse_mx6.hip): VFM_ABL_NOFOLD, _NOLDS, _NODMA, _NOBAR, _NOEMIT, _NOSCALE, _NOCMP, _NOTAB, _EMITSB
set -e
cd "$(dirname "$0")/../vfm-registration_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
for spec in "$@"; do
  v="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc $F $defs -c csrc/match_coarse_mx6.hip -o build/match_coarse_mx6_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vfmreg/lib/libvfmreg_hip_$v.so build/error.cpp.o build/config.cpp.o build/match_api.hip.o build/match_prep.hip.o \
        build/match_coarse_f16.hip.o build/match_coarse_i8.hip.o build/match_coarse_mx6_$v.o build/match_finish.hip.o build/match_l2.hip.o build/ransac.hip.o \
        build/project.hip.o build/vit.hip.o build/vit_mlp.hip.o build/icp.hip.o build/voxel.hip.o
done
