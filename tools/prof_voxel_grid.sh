#!/bin/bash
# kernel durations of the one-launch VoxelDownsample kernel (tools/ab_voxel_grid.py under rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_voxel_grid
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_voxel_grid -o vox -- python $R/tools/ab_voxel_grid.py > $R/gpurun_out/prof_voxel_grid.log 2>&1
python - <<'P'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = glob.glob(R + "/gpurun_out/prof_voxel_grid/**/vox_kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "voxel_robin_grid_kernel" in r["Kernel_Name"]:
        by[int(r["Grid_Size_X"]) // 256].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g in sorted(by):
    v = sorted(by[g])
    print(f"voxel_robin_grid_kernel, {g:4d} workgroups: {len(v):4d} launches, median {v[len(v)//2]:7.1f} us, min {v[0]:7.1f}")
P
