# Round-4 evidence run: GPU tests, smoke, bench (default, serial, under rocprofv3), PMC passes of the coarse kernel (fp6 half-width
# with the fused bound test = record kind 8, what the default bench runs; fp6 full width = kind 5), the fp6 kernel alone, A/B of the
# pipeline per record kind, ViT (one scan, batches, kernel by kernel), the reference-shaped API, C3 as a pipeline, other rows, soaks.
#   -> gpurun_out/r04final/, collected by tools/refresh_profiles_r04.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_streams1.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-extra > $O/bench_prof.json 2> $O/prof.err; tail -1 $O/bench_prof.json | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o bench1 -- python $R/bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_prof1.json 2> $O/prof1.err
cd $R && VFM_RECORDS=8 bash tools/pmc_coarse.sh 2>&1 | tail -22
cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/pmc_match_coarse_mx6half.json 2>/dev/null
for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_mx6half_pass${i}_counter_collection.csv 2>/dev/null; done
cd $R && VFM_RECORDS=5 bash tools/pmc_coarse.sh 2>&1 | tail -22
cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/pmc_match_coarse_mx6.json 2>/dev/null
for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_mx6_pass${i}_counter_collection.csv 2>/dev/null; done
cd $R
timeout 600 python tools/dev_mx6.py > $O/dev_mx6.txt 2>&1; tail -8 $O/dev_mx6.txt
timeout 900 python tools/ab_r4.py > $O/ab_r4.txt 2>&1; tail -12 $O/ab_r4.txt
timeout 600 python tools/soak_mx6.py 40 303 2>&1 | tail -3 > $O/soak_mx6.txt; cat $O/soak_mx6.txt
timeout 600 python tools/sweep_slices_r4.py > $O/sweep_slices.txt 2>&1
{ bash tools/trace_pipe.sh mx6-half 2>&1 | tail -40; } > $O/pipeline_cycle.txt
# ViT: one scan and batches, kernel by kernel
timeout 600 python tools/time_vit_batch.py > $O/time_vit_batch.txt 2>&1; tail -12 $O/time_vit_batch.txt
bash tools/prof_vit_batch.sh > $O/prof_vit_batch.txt 2>&1
# the reference-shaped API (map kept between scans; with and without ICP), C3 as a pipeline
timeout 600 python tools/time_api.py > $O/time_api.txt 2>&1; tail -12 $O/time_api.txt
timeout 600 python tools/time_c3_pipe.py 0 > $O/time_c3_pipe.txt 2>&1; tail -6 $O/time_c3_pipe.txt
bash tools/trace_c3_pipe.sh > $O/trace_c3_pipe.txt 2>&1
timeout 400 python tools/time_c3_group.py 1 2 4 8 1 4 2>&1 | grep -v amdgpu > $O/time_c3_group.txt; cat $O/time_c3_group.txt
VFM_AB_IMAGES=48,72,84,90,93,96,144 timeout 400 python tools/ab_vit_astat.py 144 2>&1 | grep -v amdgpu > $O/ab_vit_astat.txt; cat $O/ab_vit_astat.txt
timeout 600 python tools/ab_prep_r4.py 2>&1 | grep -v amdgpu > $O/ab_prep_forms.txt
timeout 300 python tools/ab_vit_split.py 2>&1 | grep -v amdgpu > $O/vit_split.txt
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/hbm_probe tools/probe/hbm_probe.hip && /tmp/hbm_probe > $O/hbm_probe.txt 2>&1
# F rows, C3 stages, RANSAC alone, row A6
{ timeout 300 python tools/time_f_rows.py 2>&1; echo; timeout 300 python tools/time_c3.py 2>&1; echo; timeout 200 python tools/time_ransac.py 2>&1; } > $O/other_rows.txt; tail -30 $O/other_rows.txt
timeout 300 python tools/time_pairs.py 6 > $O/time_pairs.txt 2>&1
timeout 300 python tools/time_c3_modes.py 2>/dev/null | tail -6 > $O/time_c3_modes.txt
# duplicate-rich maps through the bench's pipeline (policy by feedback and every mode forced)
timeout 1200 python tools/time_neardup.py --steps 20 --modes auto,mx6-half,int8-half,int8,mx6 --out $O/neardup.json > $O/neardup.log 2>&1
timeout 900 python tools/soak_half.py 40 303 2>&1 | tail -3 > $O/soak_half.txt; cat $O/soak_half.txt
# the finish stage kernel by kernel on lifted descriptors with a common component (fp6 and int8 best-score records), and on D.2 (fused half-width)
{ bash tools/prof_finish.sh 5,0 50 lifted 2>&1 | tail -3; bash tools/prof_finish.sh 8 50 d2 2>&1 | tail -1; } > $O/prof_finish.txt
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/launch_probe tools/probe/launch_probe.hip && /tmp/launch_probe > $O/launch_probe.txt 2>&1
