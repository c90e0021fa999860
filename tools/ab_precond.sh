# The 20-step timed region against what precedes it: --warmup 1 / 3 / 5 with the policy's settle phase, and -- A/B
# only, not what bench.py does by default -- 40 / 200 extra untimed registrations in front of the warm-up (VFM_BENCH_PRECOND);
# per-step HIP-event durations of the coarse kernel inside the timed region.
R=$GRAFT_REPO_ROOT
cd $R
for cfg in "1 0" "3 0" "3 0" "5 0" "3 40" "3 40" "3 200"; do
  set -- $cfg
  echo "--warmup $1, VFM_BENCH_PRECOND=$2"
  VFM_BENCH_PRECOND=$2 VFM_BENCH_TRACE=1 python bench.py --steps 20 --warmup $1 --no-extra --no-cpu-baseline 2>&1 >/dev/null | grep -E "coarse kernel ms|rank 0"
done
