# Hardware-queue binding of the pipeline's side streams: bench.py plain and under torch.distributed.run (world 1), with the streams bound at
# creation (default) and lazily (VFM_LAZY_STREAMS=1), same box.  -> gpurun_out/ab_queue_touch.txt
R=$GRAFT_REPO_ROOT
cd $R
for lazy in 0 1 0 1; do
  for form in plain torchrun; do
    if [ $form = plain ]; then
      v=$(VFM_LAZY_STREAMS=$lazy timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
    else
      v=$(VFM_LAZY_STREAMS=$lazy timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --no-extra --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1))")
    fi
    echo "lazy binding $lazy, $form: $v registrations/s"
  done
done
