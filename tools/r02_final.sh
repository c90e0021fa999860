# Round-2 evidence run: GPU tests, smoke, bench (default int8 pass, fp16 pass on the same box, serial, under rocprofv3),
# PMC passes of the coarse kernel, duplicate-rich maps.  -> gpurun_out/r02final/, collected by tools/refresh_profiles_r02.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02final
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_streams1.json 2>> $O/bench.err
VFM_VARIANT=5 timeout 600 python bench.py --no-cpu-baseline --no-extra > $O/bench_f16_same_box.json 2>> $O/bench.err; tail -1 $O/bench_f16_same_box.json | cut -c1-200
VFM_COARSE=int8 timeout 600 python bench.py --no-cpu-baseline --no-extra > $O/bench_int8_full_same_box.json 2>> $O/bench.err; tail -1 $O/bench_int8_full_same_box.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-extra > $O/bench_prof.json 2> $O/prof.err; tail -1 $O/bench_prof.json | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o bench1 -- python $R/bench.py --streams 1 --no-cpu-baseline --no-extra > $O/bench_prof1.json 2> $O/prof1.err
# PMC passes: the half-width kernel the bench runs on D.2 data (VFM_RECORDS=3), then the full-width best-score kernel (0)
cd $R && VFM_RECORDS=3 bash tools/pmc_coarse.sh 2>&1 | tail -22
cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/pmc_match_coarse_half.json 2>/dev/null
for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_half_pass${i}_counter_collection.csv 2>/dev/null; done
cd $R && VFM_RECORDS=0 bash tools/pmc_coarse.sh 2>&1 | tail -22
cp $R/gpurun_out/pmc_coarse/pmc_match_coarse.json $O/ 2>/dev/null
for i in 1 2 3 4 5 6 7; do cp $R/gpurun_out/pmc_coarse/p${i}_counter_collection.csv $O/pmc_pass${i}_counter_collection.csv 2>/dev/null; done
cd $R && python tools/time_neardup.py --steps 20 --out $O/neardup.json > $O/neardup.log 2>&1
python tools/time_prep.py > $O/time_prep.txt 2>&1; cat $O/time_prep.txt
# second half of round 2: the timed region against its length, power / clock under load, ViT batch scaling, ungated routing,
# C3 per coarse mode, what each side stage costs the steady state, CU-masked side streams
bash tools/steps_sweep.sh > /dev/null 2>&1; cp $R/gpurun_out/steps_sweep.txt $O/steps_sweep.txt
bash tools/power_probe.sh > /dev/null 2>&1; cp $R/gpurun_out/power_probe.txt $O/power_probe.txt
timeout 300 python tools/time_vit_batch.py > $O/time_vit_batch.txt 2>/dev/null
{ for s in "20000 200000 384" "2000 200000 384" "300 50000 384" "20000 50000 256" "50000 1000000 768"; do echo "## $s"; timeout 200 python tools/time_ungated.py $s 2>/dev/null; done; } > $O/time_ungated.txt
timeout 300 python tools/time_c3_modes.py 2>/dev/null | tail -5 > $O/time_c3_modes.txt
# (tools/tax_probe.py predates the half-width probe of the pipeline: its no-op patching stalls there; run it with VFM_COARSE-style fixed modes only)
{ for c in "0 0" "64 0" "0 64" "64 64" "0 0"; do timeout 100 python tools/cu_mask_probe.py $c 2>/dev/null; done; } > $O/cu_mask_probe.txt
VFM_GATE=0.7999999 bash tools/prof_i8.sh 0 > $O/prof_search_c2.txt 2>&1
{ for s in "20000 200000 384" "50000 1000000 768" "20000 50000 256" "3000 100000 384"; do echo "## $s"; VFM_AB_RECORDS=0,3,1,0,3 timeout 200 python tools/ab_half.py $s 2>/dev/null; done; } > $O/ab_half.txt
bash tools/prof_c3.sh > $O/prof_c3.txt 2>&1
tail -3 $O/steps_sweep.txt | cut -c1-200; cat $O/time_c3_modes.txt
