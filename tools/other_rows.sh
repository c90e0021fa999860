# raw numbers for the non-headline rows quoted in DESIGN.md -> gpurun_out/other_rows.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/other_rows.txt
: > $O
for t in time_library_gemm.py time_c5.py time_l2.py time_c3.py time_f_rows.py time_small.py; do
  echo "## python tools/$t" >> $O
  timeout 600 python $R/tools/$t 2>&1 | grep -v "amdgpu.ids" >> $O
  echo >> $O
done
cat $O
