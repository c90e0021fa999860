# samples rocm-smi power / clocks while the registration bench runs -> gpurun_out/power_probe.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/power_probe.txt
{
echo "## idle"; rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -iE "power|sclk|mclk|fclk" | head -8
timeout 120 python $R/bench.py --steps 20000 --warmup 5 --no-cpu-baseline --no-extra > /tmp/bench_bg.json 2>/dev/null &
BPID=$!
sleep 7
for i in 1 2 3 4; do echo "## under load, sample $i"; rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "power|sclk|mclk" | head -5; sleep 1.5; done
wait $BPID
echo "## bench line of that run"; tail -1 /tmp/bench_bg.json | cut -c1-260
} > $O 2>&1
cat $O
