# Round-6 evidence run: GPU tests, smoke, bench (default with every extra, serial, under rocprofv3), PMC passes of the coarse kernels
# (record kinds 8 = headline, 5 = full-width records), the operand preparation (HBM traffic per form, A/Bs), ViT batch times, the
# reference-shaped API cold / warm / handle, C3 in groups, other rows, soaks.
#   -> gpurun_out/r06final/, collected by tools/refresh_profiles_r05.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06final
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_streams1.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-extra > $O/bench_prof.json 2> $O/prof.err; tail -1 $O/bench_prof.json | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o bench1 -- python $R/bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_prof1.json 2> $O/prof1.err
for rk in 8 5; do
  name=$( [ $rk = 8 ] && echo mx6half || ( [ $rk = 5 ] && echo mx6 || echo mx6fused ) )
  cd $R && VFM_RECORDS=$rk bash tools/pmc_coarse.sh 2>&1 | tail -22
  cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/pmc_match_coarse_$name.json 2>/dev/null
  for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_${name}_pass${i}_counter_collection.csv 2>/dev/null; done
done
cd $R
# ViT
VFM_VIT_LDS_THR=256 timeout 400 python tools/time_vit_batch.py 2>&1 | grep -v amdgpu > $O/time_vit_batch.txt; tail -7 $O/time_vit_batch.txt
timeout 500 python tools/ab_vit_fused_qkv.py 6 24 30 42 44 48 60 84 85 86 90 96 126 2>&1 | grep -v amdgpu > $O/ab_vit_fused_qkv_sweep.txt; tail -5 $O/ab_vit_fused_qkv_sweep.txt
timeout 300 python tools/trace_vit_fused.py 42 84 2>&1 | grep -v amdgpu > $O/trace_vit_fused.txt
bash tools/prof_vit_r06.sh 84 > $O/prof_vit_84images.txt 2>&1; tail -26 $O/prof_vit_84images.txt | head -8
VIT_IMAGES=84 bash tools/pmc_vit.sh > $O/pmc_vit_84.log 2>&1; cp $R/gpurun_out/pmc_vit/summary.json $O/pmc_vit_84images.json
cd $R
# operand preparation: HBM traffic of the three forms, the one-read form's checks and times, the A/Bs of the round
cd $R && bash tools/pmc_prep.sh 2>&1 | tail -8; cp gpurun_out/pmc_prep/summary.txt $O/pmc_prep.txt
timeout 300 python tools/dev_prep_once.py 2>&1 | grep -v amdgpu > $O/dev_prep_once.txt; tail -7 $O/dev_prep_once.txt
timeout 400 python tools/ab_prep_r6.py mx6-half "mx6@lifted" 2>&1 | grep -v amdgpu > $O/ab_prep_forms.txt; tail -6 $O/ab_prep_forms.txt
timeout 300 python tools/sweep_slices_r6.py 0 28 47 56 60 2>&1 | grep -v amdgpu > $O/sweep_slices_alone.txt
timeout 300 python tools/time_api_cold.py 2>&1 | grep -v amdgpu > $O/time_api_cold.txt; tail -4 $O/time_api_cold.txt
timeout 300 bash tools/trace_pipe.sh mx6-half d2 True 2>&1 | tail -45 > $O/trace_pipe_d2.txt
# the reference-shaped API, C3 in groups, other rows
timeout 600 python tools/time_api.py > $O/time_api.txt 2>&1; tail -8 $O/time_api.txt
{ timeout 200 python tools/time_api_steps.py 2>&1; timeout 200 python tools/time_api_steps.py 60000 200000 2>&1; } | grep -v amdgpu > $O/time_api_steps.txt
timeout 400 python tools/time_c3_group.py 1 2 4 8 1 4 2>&1 | grep -v amdgpu > $O/time_c3_group.txt; cat $O/time_c3_group.txt
{ timeout 300 python tools/time_f_rows.py 2>&1; echo; timeout 300 python tools/time_c3.py 2>&1; echo; timeout 200 python tools/time_ransac.py 2>&1; } > $O/other_rows.txt; tail -20 $O/other_rows.txt
{ bash tools/prof_finish.sh 5,0 50 lifted 2>&1 | tail -3; bash tools/prof_finish.sh 8 50 d2 2>&1 | tail -1; } > $O/prof_finish.txt
# soaks beyond the suite's fixed seeds (the fp6 trial covers record kind 10)
timeout 600 python tools/soak_mx6.py 40 505 2>&1 | tail -3 > $O/soak_mx6.txt; cat $O/soak_mx6.txt
timeout 600 python tools/soak_half.py 40 505 2>&1 | tail -3 > $O/soak_half.txt; cat $O/soak_half.txt
timeout 900 python tools/soak_vit_fused.py 40 606 2>&1 | grep -v amdgpu | tail -4 > $O/soak_vit_fused.txt; cat $O/soak_vit_fused.txt
