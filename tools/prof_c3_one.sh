# kernel sequence of one C3 registration (lifted ViT descriptors), mode $1 (default int8-top2) -> gpurun_out/prof_c3one
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c3one
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C3_MODES=${1:-int8-top2} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c -- python $R/tools/time_c3_modes.py > $O/out.txt 2> $O/err.txt
tail -2 $O/out.txt
python $R/tools/trace_c3.py $(find $O -name c_kernel_trace.csv | head -1)
