# Round-3 evidence run: GPU tests, smoke, bench (default, full-width int8 on the same box, serial, under rocprofv3), PMC passes
# of the coarse kernel (half-width, full-width), duplicate-rich maps, row A6 (mutual pairs), F rows, C3 stages, RANSAC,
# soaks, the fp6 MFMA probe.   -> gpurun_out/r03final/, collected by tools/refresh_profiles_r03.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03final
rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_streams1.json 2>> $O/bench.err
VFM_COARSE=int8 timeout 600 python bench.py --no-cpu-baseline --no-extra > $O/bench_int8_full_same_box.json 2>> $O/bench.err; tail -1 $O/bench_int8_full_same_box.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-extra > $O/bench_prof.json 2> $O/prof.err; tail -1 $O/bench_prof.json | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o bench1 -- python $R/bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_prof1.json 2> $O/prof1.err
cd $R && VFM_RECORDS=3 bash tools/pmc_coarse.sh 2>&1 | tail -22
cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/pmc_match_coarse_half.json 2>/dev/null
for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_half_pass${i}_counter_collection.csv 2>/dev/null; done
cd $R && VFM_RECORDS=0 bash tools/pmc_coarse.sh 2>&1 | tail -22
cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/ 2>/dev/null
for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_pass${i}_counter_collection.csv 2>/dev/null; done
cd $R && VFM_RECORDS=5 bash tools/pmc_coarse.sh 2>&1 | tail -22
cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/pmc_match_coarse_mx6.json 2>/dev/null
for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_mx6_pass${i}_counter_collection.csv 2>/dev/null; done
cd $R && VFM_RECORDS=7 bash tools/pmc_coarse.sh 2>&1 | tail -22
cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/pmc_match_coarse_mx6half.json 2>/dev/null
for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_mx6half_pass${i}_counter_collection.csv 2>/dev/null; done
timeout 600 python tools/dev_mx6.py > $O/dev_mx6.txt 2>&1; tail -8 $O/dev_mx6.txt
timeout 600 python tools/queue_probe.py int8-half private > $O/queue_probe.txt 2>&1; timeout 600 python tools/queue_probe.py int8-half shared >> $O/queue_probe.txt 2>&1; timeout 600 python tools/queue_probe.py mx6-half shared >> $O/queue_probe.txt 2>&1
timeout 900 python tools/ab_mx6_bench.py > $O/ab_mx6_bench.txt 2>&1
timeout 600 python tools/soak_mx6.py 40 303 2>&1 | tail -3 > $O/soak_mx6.txt; cat $O/soak_mx6.txt
# what a cycle of the pipeline consists of: per-stream kernel timeline (int8 half-width, fp6 half-width), thin kernels beside the coarse kernels
{ bash tools/trace_pipe.sh int8-half 2>&1 | tail -40; echo; bash tools/trace_pipe.sh mx6-half 2>&1 | tail -40; echo; timeout 300 python tools/corun_probe.py 2>&1 | tail -4; } > $O/pipeline_cycle.txt
cd $R && timeout 900 python tools/time_neardup.py --steps 20 --out $O/neardup.json > $O/neardup.log 2>&1
# the finish stage on descriptors that look like lifted ones: candidate / hit counts, kernel by kernel (alone), and the pipeline's timeline
{ timeout 300 python tools/lifted_stats.py 1.0 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/lifted_stats.py 0.0 2>&1 | grep -v amdgpu.ids; } > $O/lifted_stats.txt
{ bash tools/prof_finish.sh 0,5 0 lifted 2>&1 | grep records; bash tools/prof_finish.sh 0,5 0 d2 2>&1 | grep records; } > $O/prof_finish.txt
{ echo "== auto, overlapped"; bash tools/trace_pipe.sh auto lifted 2>&1 | tail -45; echo "== int8, every kernel alone"; bash tools/trace_pipe.sh int8 lifted False 2>&1 | tail -32; echo "== mx6, every kernel alone"; bash tools/trace_pipe.sh mx6 lifted False 2>&1 | tail -32; } > $O/lifted_cycle.txt
# the 20-step timed region against the length of what precedes it
bash tools/ab_precond.sh > $O/warmup_ab.txt 2>&1
# row A6 (find_correspondences' mutual filter): timing, kernel sequence
timeout 300 python tools/time_pairs.py 6 > $O/time_pairs.txt 2>&1; cat $O/time_pairs.txt
bash tools/prof_pairs.sh > $O/prof_pairs.txt 2>&1
# F rows, C3 stages, ViT tile mapping A/B, RANSAC alone
{ timeout 300 python tools/time_f_rows.py 2>&1; echo; timeout 300 python tools/time_c3.py 2>&1; echo; timeout 200 python tools/time_ransac.py 2>&1; } > $O/other_rows.txt; cat $O/other_rows.txt | tail -30
timeout 300 python tools/time_c3_modes.py 2>/dev/null | tail -6 > $O/time_c3_modes.txt
timeout 300 python tools/ab_vit_xcd.py > $O/ab_vit_xcd.txt 2>&1
bash tools/prof_c3_one.sh > $O/prof_c3_one.txt 2>&1
# soaks beyond the suite's fixed seeds
timeout 900 python tools/soak_half.py 40 303 2>&1 | tail -3 > $O/soak_half.txt; cat $O/soak_half.txt
timeout 900 python tools/soak_match.py 16 303 2>&1 | tail -3 > $O/soak_match.txt; cat $O/soak_match.txt
# fp6 (MX e2m3) MFMA: operand layout check + issue rate
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mx6_probe tools/probe/mx6_probe.hip && /tmp/mx6_probe > $O/mx6_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mx6_cvt_probe tools/probe/mx6_cvt_probe.hip && /tmp/mx6_cvt_probe >> $O/mx6_probe.txt 2>&1; cat $O/mx6_probe.txt
