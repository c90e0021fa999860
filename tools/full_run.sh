R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err; tail -1 $O/bench_prof.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o bench1 -- python $R/bench.py --streams 1 --no-cpu-baseline > $O/bench_prof1.json 2> $O/prof1.err; tail -1 $O/bench_prof1.json
