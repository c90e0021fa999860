# End-of-round refresh of what changed after tools/r04_final.sh's run (the kernels did not): GPU tests, smoke, the default bench (with the
# grouped C3 pipeline, the stage table and the traffic measured in the run), the serial bench, both under rocprofv3 --stats, C3 grouped.
#   -> gpurun_out/r04final/, collected by tools/refresh_profiles_r04.py (files it does not find keep their earlier versions under profiles/)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_streams1.json 2>> $O/bench.err
timeout 400 python tools/time_c3_group.py 1 2 4 8 1 4 2>&1 | grep -v amdgpu > $O/time_c3_group.txt; cat $O/time_c3_group.txt
VFM_VIT_LDS_THR=256 timeout 400 python tools/time_vit_batch.py 2>&1 | grep -v amdgpu > $O/time_vit_batch.txt; tail -7 $O/time_vit_batch.txt
VFM_AB_IMAGES=48,72,84,90,93,96,144 timeout 400 python tools/ab_vit_astat.py 144 2>&1 | grep -v amdgpu > $O/ab_vit_astat.txt; cat $O/ab_vit_astat.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-extra > $O/bench_prof.json 2> $O/prof.err; tail -1 $O/bench_prof.json | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o bench1 -- python $R/bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_prof1.json 2> $O/prof1.err
