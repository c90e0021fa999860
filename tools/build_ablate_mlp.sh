#!/bin/bash
# timing-experiment builds of vit_mlp_kernel next to the real library (results are garbage):
#   tools/build_ablate_mlp.sh NAME:"-DVFM_MLP_ABL_NOGELU" ...  ->  vfmreg/lib/libvfmreg_hip_NAME.so     switches: VFM_MLP_ABL_NOGELU, _NOLDS, _NODMA
set -e
cd "$(dirname "$0")/../vfm-registration_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DVFM_VIT_PART=1"
for spec in "$@"; do
  v="${spec%%:*}"; defs="${spec#*:}"
  [ "$defs" = "$spec" ] && defs=""
  /opt/rocm/bin/hipcc $F $defs -c csrc/vit.hip -o build/vit_mlp_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vfmreg/lib/libvfmreg_hip_$v.so build/error.cpp.o build/config.cpp.o build/match_api.hip.o build/match_prep.hip.o \
        build/match_coarse_f16.hip.o build/match_coarse_i8.hip.o build/match_coarse_mx6.hip.o build/match_finish.hip.o build/match_l2.hip.o build/ransac.hip.o \
        build/project.hip.o build/vit.hip.o build/vit_mlp_$v.o build/icp.hip.o build/voxel.hip.o
done
