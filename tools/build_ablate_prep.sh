#!/bin/bash
# builds timing-experiment variants of the operand-preparation kernel next to the real library (results of such builds are garbage):
#   tools/build_ablate_prep.sh NAME:"-DVFM_PABL_NOST8 -DVFM_PABL_NOCONV" ...   ->  vfmreg/lib/libvfmreg_hip_NAME.so
# switches (csrc/match_prep.hip): VFM_PABL_NOLOAD, _NOP2, _NOST8, _NOMX6, _NOCONV, _NOST6
set -e
cd "$(dirname "$0")/../vfm-registration_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
for spec in "$@"; do
  v="${spec%%:*}"; defs="${spec#*:}"
  [ "$defs" = "$spec" ] && defs=""
  /opt/rocm/bin/hipcc $F $defs -c csrc/match_prep.hip -o build/match_prep_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vfmreg/lib/libvfmreg_hip_$v.so build/error.cpp.o build/config.cpp.o build/match_api.hip.o build/match_prep_$v.o \
        build/match_coarse_f16.hip.o build/match_coarse_i8.hip.o build/match_coarse_mx6.hip.o build/match_finish.hip.o build/match_l2.hip.o build/ransac.hip.o \
        build/project.hip.o build/vit.hip.o build/vit_mlp.hip.o build/icp.hip.o build/voxel.hip.o
done
